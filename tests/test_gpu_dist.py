"""The multi-GPU schedules INSIDE the library (csrc/abi_dist.hip: hodor_dist_ntt_forward / _inverse / _begin / _end /
_natural, hodor_dist_lde_by_cosets, hodor_dist_commit) — what a Rust process per GPU binds instead of writing a schedule
of its own.  The box has one GPU, so:
  * world 1, every transport (the direct stores, the copy engine, RCCL on a real one-rank communicator) with the
    exchanges FORCED: the whole call sequence, flags and streams included, against the single-device transform / LDE /
    tree and the CPU oracle's committed digests;
  * ranks played in ONE process, each on a stream of its own, through the split-phase pair (begin for every rank, then
    end for every rank — what two GPUs do concurrently): every rank's block against the single-device transform;
  * two and four PROCESSES sharing the GPU over hipIpc handles (tests/dist_worker.py): natural-order transforms, two
    transforms in flight, LDE by cosets + commit by subtrees in both tree formats, every rank checking its share.
The transport between two real devices stays unexercised, as everywhere in SURVEY §8(e)."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")))


def _handle(ctx, kind, n_local):
    import hodor_amd
    from hodor_amd import _lib
    if kind == "rccl":
        if not hodor_amd.Exchange.available():
            pytest.skip("librccl cannot be bound in this process")
        x = hodor_amd.Exchange(ctx, hodor_amd.Exchange.unique_id(), 1, 0)
        x.set_transport(_lib.RCCL, force_collectives=True)
        return x
    x = hodor_amd.DirectExchange(ctx, 1, 0, n_local, n_slots=4)
    hodor_amd.DirectExchange.connect_local([x])
    x.set_transport(_lib.COPY if kind == "copy" else _lib.DIRECT, force_collectives=True)
    return x


@pytest.mark.parametrize("kind", ["direct", "copy", "rccl"])
@pytest.mark.parametrize("log_n,log_chunks", [(8, 0), (13, 1), (20, 2)])
def test_dist_transforms_at_world_1_over_every_transport(gpu_ctxs, kind, log_n, log_chunks):
    import torch
    ctx = gpu_ctxs["bn256"]
    n = 1 << log_n
    x = _handle(ctx, kind, n)
    a = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    ctx.gen_elements_dev(a, 0, n, 4000 + log_n)
    w = ctx.domain(n)[2]
    spec = torch.empty_like(a)
    ctx.poly_fft_dev(a, spec, log_n)
    for _ in range(3):                      # slots and work buffers come round again
        b = x.dist_forward(a, torch.empty_like(a), log_n, w, log_chunks)
        back = x.dist_inverse(b, torch.empty_like(a), log_n, w, log_chunks)
        nat = x.dist_natural(a, torch.empty_like(a), log_n, w, False)
        inv = x.dist_natural(nat, torch.empty_like(a), log_n, w, True)
    ctx.synchronize()
    l1 = 9 if 19 <= log_n <= 27 else log_n // 2
    assert torch.equal(b.view(1 << l1, 1 << (log_n - l1), 4).permute(1, 0, 2).contiguous().view(-1, 4), spec)   # B = X[k1 + N1 k2]
    assert torch.equal(back, a) and torch.equal(nat, spec) and torch.equal(inv, a)
    x.close()


@pytest.mark.parametrize("kind", ["direct", "copy", "rccl"])
def test_dist_lde_and_commit_at_world_1_match_the_cpu_oracle(gpu_ctxs, oracles, kind):
    """hodor_dist_lde_by_cosets_dev + hodor_dist_commit_dev at world 1 with the exchanges forced: the 2^18-coefficient
    LDE x8 and its tree hash to the CPU oracle's committed digests (tests/golden/fullsize_digests.json); coset and
    paired (COSET2) variants against the single-device calls."""
    import torch
    import hodor_amd
    ctx = gpu_ctxs["bn256"]
    fx = FULL["lde"]["18"]
    log_n, factor = 18, fx["factor"]
    n, big = 1 << log_n, (1 << log_n) * factor
    x = _handle(ctx, kind, big)
    coeffs = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    ctx.gen_elements_dev(coeffs, 0, n, fx["seed"])
    blk = x.dist_lde_by_cosets(coeffs, log_n, factor, torch.empty((big, 4), dtype=torch.int64, device="cuda"))
    nodes = torch.empty((big, 32), dtype=torch.uint8, device="cuda")
    root, top = x.dist_commit(blk, nodes, hodor_amd.TRIVIAL)
    ctx.synchronize()
    assert hashlib.blake2s(memoryview(blk.cpu().numpy()).cast("B"), digest_size=32).hexdigest() == fx["lde"]
    assert root.hex() == fx["root"] and top[1] == root
    for coset in (False, True):
        ref = torch.empty((big, 4), dtype=torch.int64, device="cuda")
        ctx.poly_lde_dev(coeffs, ref, log_n, factor, coset=coset)
        got = x.dist_lde_by_cosets(coeffs, log_n, factor, torch.empty_like(ref), coset=coset)
        paired = x.dist_lde_by_cosets(coeffs, log_n, factor, torch.empty_like(ref), coset=coset, paired=True)
        nodes2 = torch.empty((big // 2, 32), dtype=torch.uint8, device="cuda")
        root2, _ = x.dist_commit(paired, nodes2, hodor_amd.COSET2)
        ref2 = torch.empty_like(nodes2)
        ctx.iop_create_combined_dev(ref, big, hodor_amd.COSET2, ref2)
        ctx.synchronize()
        assert torch.equal(got, ref) and torch.equal(paired, ref)      # one rank: the paired block IS the natural order
        assert torch.equal(nodes2, ref2) and root2 == bytes(ref2[1].cpu().numpy())
    x.close()


@pytest.mark.parametrize("kind", ["direct", "copy"])
@pytest.mark.parametrize("world,log_n,log_chunks", [(2, 10, 0), (2, 14, 1), (4, 12, 2), (8, 16, 1), (2, 20, 2)])
def test_dist_split_phase_with_ranks_played_on_streams_of_their_own(gpu_ctxs, kind, world, log_n, log_chunks):
    """Every rank's hodor_dist_ntt_begin_dev, then every rank's hodor_dist_ntt_end_dev, each rank on its own stream of the
    one device: the producers store into / the copy engine fills the other ranks' receive buffers while their wait
    kernels poll — the protocol of a node, minus the wire."""
    import torch
    import hodor_amd
    from hodor_amd import _lib
    from sixstep_ref import layout_a_torch, layout_b_torch
    ctx = gpu_ctxs["bn256"]
    n = 1 << log_n
    m = n // world
    full = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    ctx.gen_elements_dev(full, 0, n, 5000 + log_n)
    w = ctx.domain(n)[2]
    spec = torch.empty_like(full)
    ctx.poly_fft_dev(full, spec, log_n)
    torch.cuda.synchronize()
    l1 = 9 if 19 <= log_n <= 27 else log_n // 2
    l2 = log_n - l1
    xs = [hodor_amd.DirectExchange(ctx, world, r, m, n_slots=2) for r in range(world)]
    hodor_amd.DirectExchange.connect_local(xs)
    streams = [torch.cuda.Stream() for _ in range(world)]
    for x in xs:
        x.set_transport(_lib.COPY if kind == "copy" else _lib.DIRECT)
    a = [layout_a_torch(full, l1, l2, r, world) for r in range(world)]
    torch.cuda.synchronize()
    for rnd in range(3):                       # three generations per slot pair: release / begin handshakes run
        hs = [xs[r].dist_begin(a[r], log_n, w, False, log_chunks, stream=streams[r].cuda_stream) for r in range(world)]
        b = [xs[r].dist_end(hs[r], torch.empty_like(a[r])) for r in range(world)]
        hs = [xs[r].dist_begin(b[r], log_n, w, True, log_chunks, stream=streams[r].cuda_stream) for r in range(world)]
        back = [xs[r].dist_end(hs[r], torch.empty_like(a[r])) for r in range(world)]
        torch.cuda.synchronize()
        for r in range(world):
            assert torch.equal(b[r], layout_b_torch(spec, l1, l2, r, world)), (rnd, r)
            assert torch.equal(back[r], a[r]), (rnd, r)
    for x in xs:
        assert ctx.L.hodor_exchange_direct_status(x.h) == 0
        x.close()


@pytest.mark.parametrize("kind", ["direct", "copy"])
def test_dist_refuses_a_transform_larger_than_the_receive_buffers(gpu_ctxs, kind):
    """The peer-mapped transports store into receive buffers the library sized at creation: a transform whose block
    does not fit is HODOR_ERR_SIZE before anything is enqueued, and the refusal leaves the handle usable (its two
    transform slots are not leaked)."""
    import torch
    import hodor_amd
    from hodor_amd import _lib
    ctx = gpu_ctxs["bn256"]
    small, big = 10, 12
    x = hodor_amd.DirectExchange(ctx, 1, 0, 1 << small, n_slots=2)
    hodor_amd.DirectExchange.connect_local([x])
    x.set_transport(_lib.COPY if kind == "copy" else _lib.DIRECT, force_collectives=True)
    a_big = torch.empty((1 << big, 4), dtype=torch.int64, device="cuda")
    ctx.gen_elements_dev(a_big, 0, 1 << big, 77)
    for _ in range(3):                           # more refusals than the handle has transform slots
        with pytest.raises(hodor_amd.HodorError) as e:
            x.dist_begin(a_big, big, ctx.domain(1 << big)[2], False, 0)
        assert e.value.code == hodor_amd.ERR_SIZE
    a = a_big[:1 << small].contiguous()
    spec = torch.empty_like(a)
    ctx.poly_fft_dev(a, spec, small)
    w = ctx.domain(1 << small)[2]
    b = x.dist_end(x.dist_begin(a, small, w, False, 0), torch.empty_like(a))
    back = x.dist_end(x.dist_begin(b, small, w, True, 0), torch.empty_like(a))
    torch.cuda.synchronize()
    from sixstep_ref import layout_b_torch
    l1 = small // 2
    assert torch.equal(b, layout_b_torch(spec, l1, small - l1, 0, 1)) and torch.equal(back, a)
    x.status()
    x.close()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("transport,world,log_n,lde_log_n,factor", [("direct", 2, 16, 12, 8), ("copy", 2, 16, 12, 8),
                                                                    ("direct", 4, 20, 14, 8), ("copy", 2, 21, 16, 16)])
def test_dist_schedules_between_processes_sharing_the_gpu(transport, world, log_n, lde_log_n, factor):
    env = {k: v for k, v in os.environ.items() if not k.startswith("HODOR_")}
    env.update({"WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())})
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), transport, str(log_n),
                                       str(lde_log_n), str(factor)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d: %s %s" % (r, so[-1500:], se[-3000:])
        line = json.loads([l for l in so.splitlines() if l.startswith("{")][-1])
        assert line["rank"] == r and all(v is True for v in line["checks"].values()), line
