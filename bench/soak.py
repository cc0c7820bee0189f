"""Determinism soak: repeat FRI commits / Merkle trees / transforms many times and require identical bytes.
Catches intermittent races (the fused kernels hand data between phases through LDS and global memory)."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hodor_amd
from inputs import random_elements
ctx = hodor_amd.Context(device=0)
bad = 0
for log_code in (4, 6, 9, 10, 11, 12, 13, 16, 19, 20, 21):
    f = 8 if log_code >= 5 else 4
    log_deg = log_code - (3 if f == 8 else 2)
    n = 1 << log_code
    coeffs = random_elements(torch, 1 << log_deg, 100 + log_code)
    code = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    ctx.poly_lde_dev(coeffs, code, log_deg, f)
    torch.cuda.synchronize()
    ref = None
    reps = 300 if log_code <= 16 else 60
    for r in range(reps):
        p = ctx.fri_commit_dev(code, n, f, 1)
        digest = hashlib.sha256(p.serialized).hexdigest()
        if r % 7 == 0:   # also the device-resident vectors and trees of a middle round
            step = p.num_steps // 2
            size = n >> (step + 1)
            digest += hashlib.sha256(p.intermediate_values(step, size).tobytes()).hexdigest()
            digest += hashlib.sha256(p.tree_nodes(step, size).tobytes()).hexdigest()
        else:
            digest += "-"
        p.free()
        key = (r % 7 == 0)
        if ref is None:
            ref = {}
        if key not in ref:
            ref[key] = digest
        elif ref[key] != digest:
            bad += 1
            print("MISMATCH fri commit 2^%d rep %d" % (log_code, r))
    print("fri commit 2^%d: %d repetitions identical" % (log_code, reps))
for log_n in (8, 12, 17, 20, 22):
    n = 1 << log_n
    a = random_elements(torch, n, 5)
    b = torch.empty_like(a)
    nodes = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    ref = None
    for r in range(100):
        ctx.poly_fft_dev(a, b, log_n)
        ctx.iop_create_dev(b, n, nodes)
        torch.cuda.synchronize()
        d = hashlib.sha256(b.cpu().numpy().tobytes()).hexdigest() + hashlib.sha256(nodes.cpu().numpy().tobytes()).hexdigest()
        if ref is None:
            ref = d
        elif ref != d:
            bad += 1
            print("MISMATCH ntt+tree 2^%d rep %d" % (log_n, r))
    print("ntt + tree 2^%d: 100 repetitions identical" % log_n)
print("SOAK", "FAILED" if bad else "OK", bad)
sys.exit(1 if bad else 0)
