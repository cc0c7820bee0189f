#!/usr/bin/env python3
"""What the start-up self-test adds to hodor_ctx_create (csrc/abi_selftest.hip): contexts created in a loop with and
without it (HODOR_SELFTEST is read once per process, hence one subprocess per setting); the first context of a process
also pays HIP's start-up and the code objects' first load and is reported apart."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r"""
import time, hodor_amd
ts = []
for i in range(12):
    t = time.perf_counter()
    ctx = hodor_amd.Context(device=0)
    ts.append((time.perf_counter() - t) * 1e3)
    ctx.close()
rest = sorted(ts[2:])
print("first %.2f ms, second %.2f ms, then median %.3f ms (min %.3f, max %.3f) over %d" % (ts[0], ts[1], rest[len(rest) // 2], rest[0], rest[-1], len(rest)))
"""
for flag in ("1", "0"):
    out = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, HODOR_SELFTEST=flag), cwd=ROOT, capture_output=True, text=True)
    print("HODOR_SELFTEST=%s  hodor_ctx_create + destroy: %s" % (flag, (out.stdout.strip().splitlines() or [out.stderr[-400:]])[-1]))
