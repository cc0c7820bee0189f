"""PCIe-inclusive timing of the slice API (host `&mut [F]` in, host out) — the literal drop-in path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hodor_amd
ctx = hodor_amd.Context(device=0)
rng = np.random.default_rng(1)
for log_n in (16, 20, 22, 24):
    n = 1 << log_n
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    ctx.poly_fft(a)                      # warm-up: tables, staging buffers
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); ctx.poly_fft(a); best = min(best, time.perf_counter() - t)
    print("slice poly_fft 2^%d: %.3f ms  (%.2e elems/s, %.1f GB/s over PCIe both ways)" %
          (log_n, best * 1e3, n / best, 2 * n * 32 / best / 1e9))
