#!/usr/bin/env python3
"""Diff the Rust-produced digests with the committed ones (same keys as tests/golden/gen_fullsize.py)."""
import json
import sys

rust, ours = (json.load(open(p)) for p in sys.argv[1:3])
bad = 0
for group in ("ntt", "lde", "fri"):
    for size, entry in rust.get(group, {}).items():
        for key, val in entry.items():
            exp = ours.get(group, {}).get(size, {}).get(key)
            if exp is not None and exp != val:
                bad += 1
                print("MISMATCH %s[%s].%s: rust %s != oracle %s" % (group, size, key, val, exp))
print("compared; %d mismatches" % bad)
sys.exit(1 if bad else 0)
