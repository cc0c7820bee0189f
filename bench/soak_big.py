"""Determinism soak at the benchmark sizes: LDE + FRI commit of 2^22 / 2^24 / 2^26 codewords, 12 times each."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hodor_amd
from inputs import random_elements
ctx = hodor_amd.Context(device=0)
bad = 0
for log_code in (22, 24, 26):
    f, log_deg = 8, log_code - 3
    n = 1 << log_code
    coeffs = random_elements(torch, 1 << log_deg, 100 + log_code)
    code = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    ref = None
    for r in range(12):
        ctx.poly_lde_dev(coeffs, code, log_deg, f)
        p = ctx.fri_commit_dev(code, n, f, 1)
        d = hashlib.sha256(p.serialized).hexdigest()
        step = 2
        size = n >> (step + 1)
        d += hashlib.sha256(p.tree_nodes(step, size).tobytes()).hexdigest()
        p.free()
        if ref is None: ref = d
        elif ref != d:
            bad += 1; print("MISMATCH", log_code, r)
    print("lde + fri commit 2^%d: 12 repetitions identical" % log_code)
print("SOAK", "FAILED" if bad else "OK")
