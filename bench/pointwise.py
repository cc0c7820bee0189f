"""Throughput of the value-form polynomial operations on 2^24 elements."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hodor_amd
from inputs import random_elements
ctx = hodor_amd.Context(device=0)
n = 1 << 24
a = random_elements(torch, n, 1); b = random_elements(torch, n, 2)
g = 7
def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
rows = [
    ("add_assign", lambda: ctx.poly_binary_dev(a, b, n, "add"), 96),
    ("mul_assign", lambda: ctx.poly_binary_dev(a, b, n, "mul"), 96),
    ("add_assign_scaled", lambda: ctx.poly_add_scaled_dev(a, b, n, 12345), 96),
    ("scale", lambda: ctx.poly_unary_dev(a, n, "scale", c=12345), 64),
    ("square", lambda: ctx.poly_unary_dev(a, n, "square"), 64),
    ("distribute_powers", lambda: ctx.distribute_powers_dev(a, n, 12345), 64),
    ("batch_inversion", lambda: ctx.poly_batch_inversion_dev(a, n), 64),
    ("evaluate_at", lambda: ctx.poly_evaluate_at_dev(a, n, 12345), 32),
]
for name, f, bytes_per in rows:
    ms = timeit(f)
    print("%-20s %7.3f ms  %6.1f G elems/s  %6.0f GB/s" % (name, ms, n / ms / 1e6, n * bytes_per / ms / 1e6))
