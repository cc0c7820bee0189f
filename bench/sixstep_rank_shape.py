#!/usr/bin/env python3
"""Local arithmetic of ONE rank of the multi-GPU 4-step transform, timed on a single GPU: rank 0's two
ABI calls per direction (columns + rows) for a transform of P * 2^log_local points, P = 1, 2, 4, 8, with the
exchange left out (the received buffer is simply the sent one, so the VALUES are meaningless — only the
time is read).  This is the compute part of `bench.py --gpus P` per step; what the RCCL all-to-all adds on
top cannot be measured on the pool's single-GPU boxes.
    python bench/sixstep_rank_shape.py [log_local] [log_n1 override]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import hodor_amd
    from hodor_amd.sixstep import HipBackend, split_logs

    log_local = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    ctx = hodor_amd.Context(hodor_amd.BN256_FR_MODULUS, hodor_amd.BN256_FR_GENERATOR, device=0)
    m = 1 << log_local
    a = torch.empty((m, 4), dtype=torch.int64, device="cuda")
    ctx.gen_elements_dev(a, 0, m, 7)
    be = HipBackend(ctx)
    base = None
    for world in (1, 2, 4, 8):
        log_p = world.bit_length() - 1
        log_n = log_local + log_p
        l1, l2 = split_logs(log_n)
        if len(sys.argv) > 2:
            l1 = int(sys.argv[2]); l2 = log_n - l1
        omega = ctx.domain(1 << log_n)[2]
        for chunks in sorted({1, 1 if world == 1 else (4 if world == 2 else 8)}):
            lc = chunks.bit_length() - 1
            step = m >> lc
            send, out = torch.empty_like(a), torch.empty_like(a)

            def one_step():
                for k in range(chunks):
                    be.columns(a, l1, l2, log_p, 0, omega, False, lc, k, out=send[k * step:(k + 1) * step])
                be.rows(send, l1, l2, log_p, 0, omega, False, lc, 0, out=out)
                for k in range(chunks):
                    be.rows(out, l1, l2, log_p, 0, omega, True, lc, k, out=send[k * step:(k + 1) * step])
                be.columns(send, l1, l2, log_p, 0, omega, True, lc, 0, out=out)

            for _ in range(20):
                one_step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 50
            e0.record()
            for _ in range(reps):
                one_step()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            if base is None:
                base = ms
            wire = m * 32 * (world - 1) / world
            print("P = %d  transform 2^%d = 2^%d x 2^%d, 2^%d points per rank, %d chunk(s): local arithmetic %.3f ms per "
                  "NTT+iNTT (%.2fx the P = 1 shape); on the wire per rank and transform %.0f MB"
                  % (world, log_n, l1, l2, log_local, chunks, ms, ms / base, wire / 1e6))
        # the direct transport (csrc/abi_exchange.hip): rank 0's producers store slab t into "rank t's" receive buffer —
        # here P local allocations stand in for the peers — un-chunked (the overlap of stores and arithmetic happens
        # inside the one launch), with the begin / signal / wait / release flag kernels of a real step
        xs = [hodor_amd.DirectExchange(ctx, world, r, m, n_slots=2) for r in range(world)]
        hodor_amd.DirectExchange.connect_local(xs)
        x0 = xs[0]

        def direct_step():
            x0.begin(0)
            x0.columns(a, 0, l1, l2, omega)
            x0.signal(0)
            for p in xs[1:]:
                p.begin(0); p.signal(0)          # the played peers only keep the generation counters moving
            x0.wait(0)
            be.rows(x0.recv[0], l1, l2, log_p, 0, omega, False, 0, 0, out=out)
            x0.release(0)
            for p in xs[1:]:
                p.wait(0); p.release(0)
            x0.begin(1)
            x0.rows(out, 1, l1, l2, omega)
            x0.signal(1)
            for p in xs[1:]:
                p.begin(1); p.signal(1)
            x0.wait(1)
            be.columns(x0.recv[1], l1, l2, log_p, 0, omega, True, 0, 0, out=send)
            x0.release(1)
            for p in xs[1:]:
                p.wait(1); p.release(1)

        for _ in range(20):
            direct_step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            direct_step()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("P = %d  the same with the DIRECT transport (producers store into %d receive buffers, no chunks, flag kernels "
              "included): %.3f ms per NTT+iNTT (%.2fx the P = 1 shape)" % (world, world, ms, ms / base))
        for p in xs:
            p.close()


if __name__ == "__main__":
    main()
