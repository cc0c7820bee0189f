"""FRI commit time against codeword size: the small sizes expose the per-round launch chain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hodor_amd
from inputs import random_elements
ctx = hodor_amd.Context(device=0)
f = 8
for log_deg in (5, 9, 11, 13, 15, 17, 19, 21, 23):
    n = (1 << log_deg) * f
    coeffs = random_elements(torch, 1 << log_deg, 4242)
    code = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    ctx.poly_lde_dev(coeffs, code, log_deg, f)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t = time.perf_counter(); p = ctx.fri_commit_dev(code, n, f, 1); dt = time.perf_counter() - t; p.free()
        best = min(best, dt)
    print(f"codeword 2^{log_deg + 3}: {log_deg} rounds, fri commit {best * 1e3:.3f} ms")
