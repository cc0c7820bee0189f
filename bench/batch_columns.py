"""All registers at once (src/prover/mod.rs:73-80): batched LDE x8 + Merkle commit against one call per column."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hodor_amd
from inputs import random_elements
ctx = hodor_amd.Context(device=0)
def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
f = 8
for log_n, cols in ((16, 32), (18, 32), (20, 16), (22, 8)):
    n = 1 << log_n
    src = random_elements(torch, n * cols, 3)
    lde = torch.empty((n * f * cols, 4), dtype=torch.int64, device="cuda")
    nodes = torch.empty((n * f * cols, 32), dtype=torch.uint8, device="cuda")
    def batched():
        ctx.poly_lde_batch_dev(src, lde, log_n, f, cols)
        ctx.iop_create_batch_dev(lde, n * f, cols, nodes)
    def one_by_one():
        for c in range(cols):
            ctx.poly_lde_dev(src[c * n:(c + 1) * n], lde[c * n * f:(c + 1) * n * f], log_n, f)
            ctx.iop_create_dev(lde[c * n * f:(c + 1) * n * f], n * f, nodes[c * n * f:(c + 1) * n * f])
    tb, to = timeit(batched), timeit(one_by_one)
    print("2^%d x %2d columns, lde x8 + commit: batched %.3f ms (%.3f per column), one by one %.3f ms; %.2e leaves/s"
          % (log_n, cols, tb, tb / cols, to, n * f * cols / tb * 1e3))
    del src, lde, nodes
