"""Times plain NTTs / LDEs of several sizes under the current HODOR_* plan knobs (tuning aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hodor_amd
from inputs import random_elements  # noqa

ctx = hodor_amd.Context(device=0)
def timeit(f, reps=20):
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
out = []
for log_n in [int(x) for x in sys.argv[1].split(",")]:
    n = 1 << log_n
    a = random_elements(torch, n, 1); b = torch.empty_like(a)
    out.append("ntt2^%d=%.3f" % (log_n, timeit(lambda: ctx.poly_fft_dev(a, b, log_n))))
    del a, b
for spec in (sys.argv[2].split(",") if len(sys.argv) > 2 else []):
    log_n, f = [int(x) for x in spec.split("x")]
    n = 1 << log_n
    a = random_elements(torch, n, 1); b = torch.empty((n * f, 4), dtype=torch.int64, device="cuda")
    out.append("lde2^%dx%d=%.3f" % (log_n, f, timeit(lambda: ctx.poly_lde_dev(a, b, log_n, f))))
    del a, b
print(os.environ.get("HODOR_MAX_LOG_R", "-"), os.environ.get("HODOR_TILE_LOG", "-"), " ".join(out))
