#!/usr/bin/env python3
"""BASELINE config[4] at full size on ONE GPU: the 2^30-point transform split over 8 ranks (4-step, layouts A
and B of hodor_amd/sixstep.py), the 8 ranks played one after the other on the single device and the all-to-all
done by hand (slab s of rank t's receive buffer = slab t of rank s's send buffer), compared element for element
with the single-device transform of the same input (itself checked against the CPU oracle's digests up to 2^24
and by its inverse and independent spot checks above, tests/test_gpu_fullsize.py).  Everything of config[4]
except the RCCL transport.  Peak memory at 2^30: ~5 x 32 GiB.
    python bench/sixstep_fullsize.py [log_total=30] [world=8] [log_chunks=3]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(ctx, log_n, world, log_chunks, verbose=True, check=None, seed=0x484F444F52):
    import torch

    from hodor_amd.sixstep import HipBackend, split_logs
    be = HipBackend(ctx)
    n = 1 << log_n
    l1, l2 = split_logs(log_n)
    n1, n2 = 1 << l1, 1 << l2
    log_p = world.bit_length() - 1
    r1, c2 = n1 // world, n2 // world
    m = n // world
    K = 1 << log_chunks
    omega = ctx.domain(n)[2]
    x = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    ctx.gen_elements_dev(x, 0, n, seed)
    y = torch.empty_like(x)
    t0 = time.perf_counter()
    ctx.poly_fft_dev(x, y, log_n)
    ctx.synchronize()
    t_direct = time.perf_counter() - t0
    if check is not None:      # e.g. output points of the single-device transform by direct evaluation on the CPU oracle
        check(x, y, omega)
    xm, ym = x.view(n1, n2, 4), y.view(n2, n1, 4)          # x[n1*N2 + n2];  X[k1 + N1*k2] = ym[k2][k1]

    def exchange(send, t):
        """rank t's receive buffer: chunk k, slab s  <-  rank s's send buffer: chunk k, slab t"""
        step = m // K
        slab = step // world
        out = torch.empty((m, 4), dtype=torch.int64, device="cuda")
        for k in range(K):
            for s in range(world):
                out[k * step + s * slab:k * step + (s + 1) * slab] = send[s][k * step + t * slab:k * step + (t + 1) * slab]
        return out

    # forward: A -> columns -> exchange -> rows -> B, compared with the direct transform
    send = []
    for q in range(world):
        a = xm[:, q * c2:(q + 1) * c2].contiguous().view(m, 4)
        buf = torch.empty_like(a)
        for k in range(K):
            be.columns(a, l1, l2, log_p, q, omega, False, log_chunks, k, out=buf[k * (m // K):(k + 1) * (m // K)])
        send.append(buf)
        del a
    ctx.synchronize()
    bs = []
    for q in range(world):
        b = be.rows(exchange(send, q), l1, l2, log_p, q, omega, False, log_chunks, 0)
        want = ym[:, q * r1:(q + 1) * r1].transpose(0, 1)   # b[i][k2] = X[(q*r1 + i) + N1*k2]
        if not torch.equal(b.view(r1, n2, 4), want):
            raise SystemExit("forward: rank %d's row block differs from the single-device transform" % q)
        bs.append(b)
    del send, y, ym
    torch.cuda.empty_cache()
    # inverse: B -> rows^-1 -> exchange -> columns^-1 -> A, compared with the input
    send = []
    for q in range(world):
        buf = torch.empty_like(bs[q])
        for k in range(K):
            be.rows(bs[q], l1, l2, log_p, q, omega, True, log_chunks, k, out=buf[k * (m // K):(k + 1) * (m // K)])
        send.append(buf)
    ctx.synchronize()
    del bs
    torch.cuda.empty_cache()
    for q in range(world):
        a2 = be.columns(exchange(send, q), l1, l2, log_p, q, omega, True, log_chunks, 0)
        if not torch.equal(a2.view(n1, c2, 4), xm[:, q * c2:(q + 1) * c2]):
            raise SystemExit("inverse: rank %d's column block differs from the input" % q)
    if verbose:
        print("2^%d points = 2^%d x 2^%d over %d ranks (played on one GPU, %d chunk(s) per exchange): every row block "
              "of the forward transform equals the single-device transform (%.0f ms incl. table build), the inverse "
              "returns the input; peak device memory %.1f GiB"
              % (log_n, l1, l2, world, K, t_direct * 1e3, torch.cuda.max_memory_allocated() / 2**30))
    return True


def main():
    import hodor_amd
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    log_chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    ctx = hodor_amd.Context(hodor_amd.BN256_FR_MODULUS, hodor_amd.BN256_FR_GENERATOR, device=0)
    run(ctx, log_n, world, log_chunks)


if __name__ == "__main__":
    main()
