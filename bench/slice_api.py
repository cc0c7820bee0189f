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

# the same with the caller's buffer pinned once (hodor_host_register): DMA at the link rate
for log_n in (20, 24):
    n = 1 << log_n
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    ctx.host_register(a)
    ctx.poly_fft(a)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); ctx.poly_fft(a); best = min(best, time.perf_counter() - t)
    ctx.host_unregister(a)
    print("slice poly_fft 2^%d, pinned buffer: %.3f ms  (%.2e elems/s, %.1f GB/s over PCIe both ways)" %
          (log_n, best * 1e3, n / best, 2 * n * 32 / best / 1e9))

# where the time of ONE call goes: the two copies alone (torch pageable / pinned host tensors over the same link) and the
# device-resident transform between them — a call can overlap neither copy with the other (every output depends on every
# input), only the transform with them
import torch
log_n = 24
n = 1 << log_n
host = torch.from_numpy(rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64).view(np.int64))
dev = torch.empty((n, 4), dtype=torch.int64, device="cuda")
out = torch.empty_like(dev)
for pin in (False, True):
    h = host.pin_memory() if pin else host
    back = torch.empty_like(h).pin_memory() if pin else torch.empty_like(h)
    ups, downs = [], []
    for _ in range(4):
        torch.cuda.synchronize(); t = time.perf_counter(); dev.copy_(h); torch.cuda.synchronize(); ups.append(time.perf_counter() - t)
        torch.cuda.synchronize(); t = time.perf_counter(); back.copy_(dev); torch.cuda.synchronize(); downs.append(time.perf_counter() - t)
    print("2^24 elements (512 MiB), %s host memory: upload %.2f ms (%.1f GB/s), download %.2f ms (%.1f GB/s)" %
          ("pinned" if pin else "pageable", min(ups) * 1e3, n * 32 / min(ups) / 1e9, min(downs) * 1e3, n * 32 / min(downs) / 1e9))
ctx.poly_fft_dev(dev, out, log_n)
ctx.synchronize()
t = time.perf_counter()
for _ in range(10):
    ctx.poly_fft_dev(dev, out, log_n)
ctx.synchronize()
print("2^24 transform, device-resident: %.2f ms" % ((time.perf_counter() - t) / 10 * 1e3))

# concurrent callers on one context (the reference calls best_fft from several scoped threads,
# src/arp/per_register/mod.rs:43-49): uploads, kernels and downloads of different callers overlap
import threading
for log_n in (20, 24):
    n = 1 << log_n
    for nthreads in (1, 2, 3, 4):
        arrays = [rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64) for _ in range(nthreads)]
        reps = 4
        def work(a):
            for _ in range(reps):
                ctx.poly_fft(a)
        for a in arrays:
            ctx.poly_fft(a)              # warm-up (first touch of every lane's staging buffers)
        ts = [threading.Thread(target=work, args=(a,)) for a in arrays]
        t = time.perf_counter()
        for th in ts: th.start()
        for th in ts: th.join()
        dt = time.perf_counter() - t
        total = nthreads * reps
        print("slice poly_fft 2^%d, %d threads: %.3f ms per transform (%.2e elems/s aggregate)" %
              (log_n, nthreads, dt / total * 1e3, total * n / dt))
