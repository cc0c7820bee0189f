import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hodor_amd
from inputs import random_elements
ctx = hodor_amd.Context(device=0)
log_deg, f = 23, 8
n = (1 << log_deg) * f
coeffs = random_elements(torch, 1 << log_deg, 4242)
code = torch.empty((n, 4), dtype=torch.int64, device="cuda")
ctx.poly_lde_dev(coeffs, code, log_deg, f)
torch.cuda.synchronize()
for _ in range(3):
    t = time.perf_counter(); p = ctx.fri_commit_dev(code, n, f, 1); dt = time.perf_counter() - t; p.free()
    print("fri commit ms", dt * 1e3)
