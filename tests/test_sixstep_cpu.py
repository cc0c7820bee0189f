"""world_size-2 (and 4) gloo test of the distributed 4-step / 6-step NTT schedule (hodor_amd/sixstep.py):
the all-to-all exchanges run for real on CPU tensors; the local steps (what the C ABI's
hodor_sixstep_*_dev entry points compute on the GPU) are restated here with numpy index maps + the CPU
oracle.  Results must equal the single-device transform, in natural order and in the A / B layouts."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pyref as P


from sixstep_ref import OracleBackend, layout_a, layout_b  # noqa: E402,F401


def _worker(rank, world, port, log_n, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hodor_amd.sixstep import sixstep_intt, sixstep_ntt
        be = OracleBackend()
        O = be.O
        n = 1 << log_n
        full = O.random_elements(n, 4321)                       # same seed on every rank
        _, k, omega = O.domain(n)
        blk = n // world
        mine = torch.from_numpy(full[rank * blk:(rank + 1) * blk].copy().view(np.int64))
        out = sixstep_ntt(be, mine, log_n, omega, rank, world)
        exp = full.copy()
        O.serial_fft(exp, omega, k)
        ok_fwd = np.array_equal(out.numpy().view(np.uint64), exp[rank * blk:(rank + 1) * blk])
        back = sixstep_intt(be, out, log_n, omega, rank, world)
        ok_inv = np.array_equal(back.numpy().view(np.uint64), full[rank * blk:(rank + 1) * blk])
        # the one-exchange forms on the A / B layouts
        from hodor_amd.sixstep import sixstep_forward, sixstep_inverse
        a = torch.from_numpy(layout_a(full, log_n, rank, world).view(np.int64))
        b = sixstep_forward(be, a, log_n, omega, rank, world)
        ok_b = np.array_equal(b.numpy().view(np.uint64), layout_b(exp, log_n, rank, world))
        a2 = sixstep_inverse(be, b, log_n, omega, rank, world)
        ok_a = np.array_equal(a2.numpy().view(np.uint64), a.numpy().view(np.uint64))
        # the exchange cut into overlapped chunks (asynchronous all-to-alls)
        for log_chunks in (1, 2):
            from hodor_amd.sixstep import split_logs
            l1, l2 = split_logs(log_n)
            if log_chunks > min(l1, l2) - (world.bit_length() - 1):
                continue
            bc = sixstep_forward(be, a, log_n, omega, rank, world, log_chunks=log_chunks)
            ok_b &= np.array_equal(bc.numpy().view(np.uint64), b.numpy().view(np.uint64))
            ac = sixstep_inverse(be, bc, log_n, omega, rank, world, log_chunks=log_chunks)
            ok_a &= np.array_equal(ac.numpy().view(np.uint64), a.numpy().view(np.uint64))
        # the split-phase form as bench.py pipelines it across steps: forward(i+1) begun, then the inverse of step i
        # begun, then both finished — two exchanges in flight at once
        from hodor_amd.sixstep import (sixstep_forward_begin, sixstep_forward_end, sixstep_inverse_begin,
                                       sixstep_inverse_end)
        pending = None
        for it in range(3):
            f = sixstep_forward_begin(be, a, log_n, omega, rank, world, log_chunks=1)
            inv = sixstep_inverse_begin(be, pending, log_n, omega, rank, world, log_chunks=1) if pending is not None else None
            bb = sixstep_forward_end(be, f)
            ok_b &= np.array_equal(bb.numpy().view(np.uint64), b.numpy().view(np.uint64))
            if inv is not None:
                ok_a &= np.array_equal(sixstep_inverse_end(be, inv).numpy().view(np.uint64), a.numpy().view(np.uint64))
            pending = bb
        ok_a &= np.array_equal(sixstep_inverse(be, pending, log_n, omega, rank, world, log_chunks=1).numpy().view(np.uint64),
                               a.numpy().view(np.uint64))
        ret[rank] = (ok_fwd, ok_inv, ok_b, ok_a)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,log_n", [(2, 6), (2, 9), (4, 8)])
def test_sixstep_matches_single_device_transform(world, log_n):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), log_n, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        assert ret[r] == (True, True, True, True), (r, ret[r])


# ---------------------------------------------------------------- distributed LDE + Merkle commit
class OracleTreeBackend:
    def __init__(self, O):
        self.O = O

    def tree(self, leafs, combiner=0):
        arr = np.ascontiguousarray(leafs.numpy().view(np.uint64))
        return torch.from_numpy(self.O.iop_create_coset2(arr) if combiner else self.O.iop_create(arr))

    def hash_node(self, left, right):
        return self.O.hash_node(left, right)


def _commit_worker(rank, world, port, log_n, factor, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hodor_amd.distributed import global_node_index, lde_commit_distributed
        be = OracleBackend()
        O = be.O
        n = 1 << log_n
        big = n * factor
        coeffs = O.random_elements(n, 777)
        padded = np.zeros((big, 4), dtype=np.uint64)
        padded[:n] = coeffs
        blk = big // world
        mine = torch.from_numpy(padded[rank * blk:(rank + 1) * blk].copy().view(np.int64))
        _, k, Omega = O.domain(big)
        lde_block, root, local_nodes, top = lde_commit_distributed(be, OracleTreeBackend(O), mine, log_n, factor,
                                                                  Omega, rank, world)
        full = O.poly_lde(coeffs, factor)                     # single-device LDE + tree
        nodes = O.iop_create(full)
        ok_lde = np.array_equal(lde_block.numpy().view(np.uint64), full[rank * blk:(rank + 1) * blk])
        ok_root = root == bytes(nodes[1])
        ok_top = all(top[g] == bytes(nodes[g]) for g in top)
        ln = local_nodes.numpy()
        ok_nodes = True
        w = blk // 2
        while w >= 1:
            for j in range(w):
                ok_nodes &= bytes(ln[w + j]) == bytes(nodes[global_node_index(w + j, w, rank, world)])
            w //= 2
        ret[rank] = (ok_lde, ok_root, ok_top, ok_nodes)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,log_n,factor", [(2, 5, 4), (4, 6, 8)])
def test_distributed_lde_commit_matches_single_device(world, log_n, factor):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_commit_worker, args=(world, _free_port(), log_n, factor, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        assert ret[r] == (True, True, True, True), (r, ret[r])


# ---------------------------------------------------------------- LDE dealt by cosets (one all-to-all) + commit
def _coset_worker(rank, world, port, log_n, factor, coset, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hodor_amd.distributed import global_node_index, lde_commit_by_cosets_distributed
        be = OracleBackend()
        O = be.O
        n = 1 << log_n
        big = n * factor
        coeffs = O.random_elements(n, 991)                    # replicated: same seed on every rank
        _, _, Omega = O.domain(big)
        shift = O.const("generator") if coset else None
        d = torch.from_numpy(coeffs.copy().view(np.int64))
        lde_block, root, local_nodes, top = lde_commit_by_cosets_distributed(
            be, OracleTreeBackend(O), d, log_n, factor, Omega, rank, world, coset_shift=shift)
        full = O.poly_lde(coeffs, factor, coset)              # lde_using_multiple_cosets / coset_lde (:418-482, :544-609)
        nodes = O.iop_create(full)
        blk = big // world
        ok_lde = np.array_equal(lde_block.numpy().view(np.uint64), full[rank * blk:(rank + 1) * blk])
        ok_root = root == bytes(nodes[1])
        ln = local_nodes.numpy()
        ok_nodes = True
        w = blk // 2
        while w >= 1:
            for j in range(w):
                ok_nodes &= bytes(ln[w + j]) == bytes(nodes[global_node_index(w + j, w, rank, world)])
            w //= 2
        ret[rank] = (ok_lde, ok_root, ok_nodes, np.array_equal(d.numpy().view(np.uint64), coeffs))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,log_n,factor,coset", [(2, 5, 4, False), (2, 4, 8, True), (4, 6, 8, False), (4, 5, 4, True)])
def test_lde_by_cosets_distributed_matches_single_device(world, log_n, factor, coset):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_coset_worker, args=(world, _free_port(), log_n, factor, coset, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        assert ret[r] == (True, True, True, True), (r, ret[r])


# ---------------------------------------------------------------- COSET2 trees across ranks: paired blocks
def _coset2_worker(rank, world, port, log_n, factor, coset, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hodor_amd.distributed import global_node_index, lde_commit_by_cosets_distributed
        be = OracleBackend()
        O = be.O
        n = 1 << log_n
        big = n * factor
        coeffs = O.random_elements(n, 1717)
        _, _, Omega = O.domain(big)
        shift = O.const("generator") if coset else None
        d = torch.from_numpy(coeffs.copy().view(np.int64))
        lde_block, root, local_nodes, top = lde_commit_by_cosets_distributed(
            be, OracleTreeBackend(O), d, log_n, factor, Omega, rank, world, coset_shift=shift, combiner=1)
        full = O.poly_lde(coeffs, factor, coset)
        nodes = O.iop_create_coset2(full)                     # the single-device COSET2 tree: big / 2 leaves
        hb = big // world // 2                                # half a block
        exp_block = np.concatenate([full[rank * hb:(rank + 1) * hb], full[big // 2 + rank * hb:big // 2 + (rank + 1) * hb]])
        ok_lde = np.array_equal(lde_block.numpy().view(np.uint64), exp_block)
        ok_root = root == bytes(nodes[1])
        ok_top = all(top[g] == bytes(nodes[g]) for g in top)
        ln = local_nodes.numpy()
        ok_nodes = ln.shape[0] == hb
        w = hb // 2
        while w >= 1:
            for j in range(w):
                ok_nodes &= bytes(ln[w + j]) == bytes(nodes[global_node_index(w + j, w, rank, world)])
            w //= 2
        ret[rank] = (ok_lde, ok_root, ok_top, ok_nodes)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,log_n,factor,coset", [(2, 5, 4, False), (2, 4, 8, True), (4, 6, 8, False), (4, 5, 4, True),
                                                      (1, 4, 4, False)])
def test_coset2_commit_by_cosets_distributed_matches_single_device(world, log_n, factor, coset):
    """The opt-in COSET2 tree format over the ranks of a node: the LDE by cosets hands out PAIRED blocks (both members of
    every combined leaf on one rank), each rank builds the COSET2 subtree of its chunk of leaves, one all-gather of the
    roots: values, every local node, the replicated top levels and the root equal the single-device COSET2 tree's."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_coset2_worker, args=(world, _free_port(), log_n, factor, coset, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        assert ret[r] == (True, True, True, True), (r, ret[r])
