"""One rank of a two- (or more-) process run of the library's own multi-GPU schedules (csrc/abi_dist.hip) with the ranks
SHARING the one GPU of the box: the processes map each other's receive buffers and flag blocks through hipIpc handles
(gloo carries the handles and nothing else) and every rank checks its share of the result against the single-device
transform / LDE / tree it computes for itself.  Started by tests/test_gpu_dist.py; prints one JSON line per rank.
    python dist_worker.py <transport: direct|copy> <log_n> <lde_log_n> <lde_factor>     (RANK / WORLD_SIZE / MASTER_* in the env)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import hodor_amd  # noqa: E402
from hodor_amd import _lib  # noqa: E402


def main():
    transport, log_n, lde_log_n, factor = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # the pool's boxes have one GPU, which the ranks share; on a node (bench/first_node.sh) every rank takes its own device
    device = int(os.environ.get("LOCAL_RANK", rank)) if os.environ.get("HODOR_DIST_DEVICE_PER_RANK") else 0
    torch.cuda.set_device(device)
    ctx = hodor_amd.Context(hodor_amd.BN256_FR_MODULUS, hodor_amd.BN256_FR_GENERATOR, device=device)
    n, P = 1 << log_n, world
    m = n // P
    big = (1 << lde_log_n) * factor
    x = hodor_amd.DirectExchange(ctx, P, rank, max(m, big // P), n_slots=4)
    x.connect_processes()
    x.set_transport(_lib.COPY if transport == "copy" else _lib.DIRECT)
    checks = {}

    # ---- one transform of 2^log_n points over the ranks: natural blocks in, natural blocks out (three exchanges)
    full = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    ctx.gen_elements_dev(full, 0, n, 777)
    w = ctx.domain(n)[2]
    spec = torch.empty_like(full)
    ctx.poly_fft_dev(full, spec, log_n)
    mine = full[rank * m:(rank + 1) * m].contiguous()
    out = x.dist_natural(mine, torch.empty_like(mine), log_n, w, False)
    back = x.dist_natural(out, torch.empty_like(out), log_n, w, True)
    torch.cuda.synchronize()
    checks["natural_forward"] = bool(torch.equal(out, spec[rank * m:(rank + 1) * m]))
    checks["natural_inverse"] = bool(torch.equal(back, mine))

    # ---- layout A -> B -> A with chunked, split-phase exchanges, two transforms in flight
    from sixstep_ref import layout_a_torch, layout_b_torch
    l1, l2 = (9, log_n - 9) if 19 <= log_n <= 27 else (log_n // 2, log_n - log_n // 2)
    a = layout_a_torch(full, l1, l2, rank, P)
    for log_chunks in (0, 2):
        h1 = x.dist_begin(a, log_n, w, False, log_chunks)
        h2 = x.dist_begin(a, log_n, w, False, log_chunks)          # a second, independent transform behind the first
        b1 = x.dist_end(h1, torch.empty_like(a))
        b2 = x.dist_end(h2, torch.empty_like(a))
        a_back = x.dist_inverse(b1, torch.empty_like(a), log_n, w, log_chunks)
        torch.cuda.synchronize()
        ok = torch.equal(b1, layout_b_torch(spec, l1, l2, rank, P)) and torch.equal(b2, b1) and torch.equal(a_back, a)
        checks["split_phase_chunks_%d" % (1 << log_chunks)] = bool(ok)

    # ---- LDE by cosets + commit by subtrees (both tree formats) against the single-device LDE and tree
    coeffs = torch.empty((1 << lde_log_n, 4), dtype=torch.int64, device="cuda")
    ctx.gen_elements_dev(coeffs, 0, 1 << lde_log_n, 778)
    B = big // P
    for coset in (False, True):
        lde = torch.empty((big, 4), dtype=torch.int64, device="cuda")
        ctx.poly_lde_dev(coeffs, lde, lde_log_n, factor, coset=coset)
        nodes = torch.empty((big, 32), dtype=torch.uint8, device="cuda")
        ctx.iop_create_dev(lde, big, nodes)
        blk = x.dist_lde_by_cosets(coeffs, lde_log_n, factor, torch.empty((B, 4), dtype=torch.int64, device="cuda"), coset=coset)
        local = torch.empty((B, 32), dtype=torch.uint8, device="cuda")
        root, top = x.dist_commit(blk, local, hodor_amd.TRIVIAL)
        torch.cuda.synchronize()
        ok = torch.equal(blk, lde[rank * B:(rank + 1) * B]) and root == bytes(nodes[1].cpu().numpy())
        ok = ok and all(top[i] == bytes(nodes[i].cpu().numpy()) for i in range(1, 2 * P))
        wl = B // 2
        while ok and wl >= 1:      # local node wl + j = global node wl P + rank wl + j
            ok = torch.equal(local[wl:2 * wl], nodes[wl * P + rank * wl:wl * P + (rank + 1) * wl])
            wl //= 2
        checks["lde_commit%s" % ("_coset" if coset else "")] = bool(ok)
        # COSET2: paired blocks, the local COSET2 tree is subtree `rank` of the global one
        nodes2 = torch.empty((big // 2, 32), dtype=torch.uint8, device="cuda")
        ctx.iop_create_combined_dev(lde, big, hodor_amd.COSET2, nodes2)
        blk2 = x.dist_lde_by_cosets(coeffs, lde_log_n, factor, torch.empty((B, 4), dtype=torch.int64, device="cuda"),
                                    coset=coset, paired=True)
        local2 = torch.empty((B // 2, 32), dtype=torch.uint8, device="cuda")
        root2, top2 = x.dist_commit(blk2, local2, hodor_amd.COSET2)
        torch.cuda.synchronize()
        half = B // 2
        exp2 = torch.cat([lde[rank * half:(rank + 1) * half], lde[big // 2 + rank * half:big // 2 + (rank + 1) * half]])
        ok2 = torch.equal(blk2, exp2) and root2 == bytes(nodes2[1].cpu().numpy())
        ok2 = ok2 and all(top2[i] == bytes(nodes2[i].cpu().numpy()) for i in range(1, 2 * P))
        checks["lde_commit_coset2%s" % ("_coset" if coset else "")] = bool(ok2)
    ctx.synchronize()
    checks["no_wait_timed_out"] = ctx.L.hodor_exchange_direct_status(x.h) == 0
    dist.barrier()
    x.close()
    ctx.close()
    print(json.dumps({"rank": rank, "world": world, "transport": transport, "checks": checks}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
