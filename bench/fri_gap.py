"""Run a few FRI commits of one size (argv[1] = log2 codeword; argv[2] = "coset2" for the COSET2 tree format) for a
rocprofv3 --kernel-trace timeline (bench/fri_gap_analyze.py summarises the last commit of the trace)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hodor_amd
from inputs import random_elements
log_code = int(sys.argv[1]) if len(sys.argv) > 1 else 16
comb = hodor_amd.COSET2 if len(sys.argv) > 2 and sys.argv[2] == "coset2" else hodor_amd.TRIVIAL
ctx = hodor_amd.Context(device=0)
f = 8
log_deg = log_code - 3
n = 1 << log_code
coeffs = random_elements(torch, 1 << log_deg, 4242)
code = torch.empty((n, 4), dtype=torch.int64, device="cuda")
ctx.poly_lde_dev(coeffs, code, log_deg, f)
torch.cuda.synchronize()
for _ in range(4):
    t = time.perf_counter(); p = ctx.fri_commit_dev(code, n, f, 1, combiner=comb); dt = time.perf_counter() - t; p.free()
    print(f"fri commit 2^{log_code}{' (COSET2)' if comb else ''}: {dt * 1e3:.3f} ms")
