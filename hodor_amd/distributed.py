"""distributed.py — the commit step over the GPUs of one node.

`merkle_commit_distributed`: rank r holds natural block r of the n leaves (exactly what
`sixstep_ntt` leaves behind).  Each rank builds the complete subtree over its block on its own GPU
(`Blake2sIopTree::create`, /root/reference/src/iop/blake2s_trivial_iop.rs:131-219); the P subtree
roots are exchanged with ONE all-gather of 32 bytes per rank (RCCL; gloo in the CPU tests) and every
rank hashes the top log2(P) levels itself, so all ranks hold the same root and the same challenge.
The global heap layout is preserved: local node `w + j` of a local level of width w is global node
`w*P + r*w + j`; the top levels (global widths < P) are replicated.

`lde_commit_distributed`: zero-padded distributed transform (`sixstep_ntt`, natural blocks in and out) followed by the distributed
commit — BASELINE config[2]'s shape across a node.

`lde_by_cosets_distributed`: the reference's own LDE schedule (`lde_using_multiple_cosets`,
src/polynomials/mod.rs:418-482: coset i of `factor` is an independent size-n transform of
coeffs * (W^i)^j) with the cosets dealt in contiguous blocks to the ranks — no communication until the
interleave `out[idx] = res[idx % f][idx / f]` (:466-479), which is ONE all-to-all (the natural-order transform route
needs three).  The coefficients (n elements, 1/f of the output) are replicated on every rank.

COSET2 trees (the opt-in format of include/hodor_gpu.h, leaf k = value[k] || value[k + N/2]) across ranks: the two members
of a leaf are N/2 apart, i.e. on ranks d and d + P/2 of the natural blocking — but in the LDE by cosets they are the SAME
coset at k and k + n/2, both still on the rank that transformed that coset when the interleaving all-to-all is issued.
`lde_by_cosets_distributed(..., paired=True)` therefore deals the k range in PAIRED blocks: rank d receives
k in [d kb/2, (d+1) kb/2) and the same range + n/2 (the same bytes on the wire, two half-size all-to-alls), i.e. the values
[d B/2, (d+1) B/2) and N/2 + the same of the natural order (B = N/P), concatenated.  Every COSET2 leaf of chunk d of the
N/2 leaves is then local, the local COSET2 tree over the block IS subtree d of the global tree, and the commit is the same
one all-gather of P roots (`merkle_commit_distributed(..., combiner=COSET2)`).
"""
import torch
import torch.distributed as dist

from .sixstep import _all_to_all, _library_transport, sixstep_ntt


class HipTreeBackend:
    """Local subtree on the MI355X through the C ABI.  `exchange`: a hodor_amd.Exchange / DirectExchange — the whole
    commit (subtree, the 32-byte exchange of the subtree roots, the replicated top levels) is then ONE library call
    (hodor_dist_commit_dev, csrc/abi_dist.hip) instead of this module's schedule over torch.distributed."""

    def __init__(self, ctx, stream=None, exchange=None):
        self.ctx, self.stream, self.exchange = ctx, stream, exchange

    def tree(self, leafs, combiner=0):
        n = leafs.shape[0]
        nodes = torch.empty((n // 2 if combiner == 1 else n, 32), dtype=torch.uint8, device=leafs.device)
        if combiner == 0:
            self.ctx.iop_create_dev(leafs, n, nodes, stream=self.stream)
        else:
            self.ctx.iop_create_combined_dev(leafs, n, combiner, nodes, stream=self.stream)
        return nodes

    def hash_node(self, left, right):
        return self.ctx.hash_node(left, right)


def merkle_commit_distributed(backend, leafs_local, rank, world, group=None, combiner=0):
    """Returns (root: bytes, local_nodes: (n/P, 32) uint8 tensor, top: dict global_node_index -> bytes).
    `top` holds the replicated levels: global node indices 1 .. 2P-1 (index P+r is rank r's subtree root).
    combiner 1 (COSET2): `leafs_local` is this rank's PAIRED block (module docstring; B >= 4 values), local_nodes has
    B/2 rows and `global_node_index` applies with the halved level widths."""
    assert world & (world - 1) == 0, "power-of-two world size"
    x = getattr(backend, "exchange", None)
    if x is not None and hasattr(x, "dist_commit"):
        n = leafs_local.shape[0]
        local_nodes = torch.empty((n // 2 if combiner == 1 else n, 32), dtype=torch.uint8, device=leafs_local.device)
        root, top = x.dist_commit(leafs_local, local_nodes, combiner, stream=backend.stream)
        return root, local_nodes, top
    if combiner:
        assert leafs_local.shape[0] >= 4, "a COSET2 subtree needs at least two leaves"
        local_nodes = backend.tree(leafs_local, combiner)
    else:
        local_nodes = backend.tree(leafs_local)
    my_root = local_nodes[1].contiguous()
    if world == 1:
        roots = [bytes(my_root.cpu().numpy())]
    else:
        if my_root.is_cuda and dist.get_backend(group) == "gloo":      # testing aid: ranks sharing one GPU (see sixstep._all_to_all)
            gathered = torch.empty((world, 32), dtype=torch.uint8)
            dist.all_gather_into_tensor(gathered, my_root.cpu().view(1, 32), group=group)
        else:
            gathered = torch.empty((world, 32), dtype=torch.uint8, device=my_root.device)
            dist.all_gather_into_tensor(gathered, my_root.view(1, 32), group=group)
        roots = [bytes(r) for r in gathered.cpu().numpy()]
    top = {world + r: roots[r] for r in range(world)}
    w = world // 2
    while w >= 1:
        for i in range(w):
            top[w + i] = backend.hash_node(top[2 * (w + i)], top[2 * (w + i) + 1])
        w //= 2
    return top[1], local_nodes, top


def global_node_index(local_index, local_width, rank, world):
    """Heap index in the n-leaf tree of local node `local_width + j` (local_index = local_width + j)."""
    j = local_index - local_width
    return local_width * world + rank * local_width + j


def lde_commit_distributed(ntt_backend, tree_backend, coeffs_block, log_n, factor, omega_big, rank, world,
                           group=None):
    """`coeffs_block`: this rank's natural block of the ZERO-PADDED coefficient vector of length
    (1 << log_n) * factor (all-zero on the ranks beyond the coefficients).  Returns
    (lde_block, root, local_nodes, top)."""
    log_big = log_n + (factor.bit_length() - 1)
    lde_block = sixstep_ntt(ntt_backend, coeffs_block, log_big, omega_big, rank, world, group)
    root, local_nodes, top = merkle_commit_distributed(tree_backend, lde_block, rank, world, group)
    return lde_block, root, local_nodes, top


def lde_by_cosets_distributed(backend, coeffs, log_n, factor, omega_big, rank, world, group=None, coset_shift=None,
                              paired=False):
    """`coeffs`: all n = 1 << log_n coefficients (replicated on every rank), shape (n, 4).  `omega_big`:
    generator W of the size n*factor domain.  `coset_shift`: g for coset_lde (values at g * W^idx), None
    for lde.  Needs world | factor and world | n.  Returns this rank's natural block of the n*factor
    values, shape (n*factor/world, 4) — what `merkle_commit_distributed` takes; with `paired` (needs 2 world | n) its
    PAIRED block instead: natural values [d B/2, (d+1) B/2) then N/2 + the same range (module docstring).

    Rank r transforms the f/P cosets i = r*f/P + t (any assignment is the reference's schedule: its cosets are
    independent work items, src/polynomials/mod.rs:446-460); every local step is a C-ABI call:
        coset i      hodor_poly_coset_fft_for_generator_dev(gen = g W^i)   distribute_powers fused into the first pass
        send slabs   hodor_sixstep_pack_dev([f/P][n] -> [P][f/P][n/P])     (nothing to do when f == P)
        exchange     ONE all-to-all of P slabs
        interleave   hodor_transpose_dev([f][n/P] -> [n/P][f])             out[(k - k0)*f + i]  (:466-479)"""
    n, f, P = 1 << log_n, factor, world
    assert f % P == 0 and n % P == 0, "world size must divide the LDE factor and the polynomial size"
    x = _library_transport(backend)
    if x is not None and (coset_shift is None or coset_shift == backend.ctx.generator):
        out = torch.empty((n * f // P, coeffs.shape[-1]), dtype=coeffs.dtype, device=coeffs.device)
        return x.dist_lde_by_cosets(coeffs, log_n, f, out, coset=coset_shift is not None, paired=paired, stream=backend.stream)
    fp, kb = f // P, n // P
    log_fp, log_p = fp.bit_length() - 1, P.bit_length() - 1
    omega = backend.pow(omega_big, f)                    # generator of the size-n domain
    res = []
    for t in range(fp):
        i = rank * fp + t
        gen = backend.pow(omega_big, i)
        if coset_shift is not None:
            gen = backend.mul(gen, coset_shift)
        if hasattr(backend, "coset_ntt"):
            res.append(backend.coset_ntt(coeffs, log_n, gen if (i != 0 or coset_shift is not None) else None, omega))
        else:
            buf = coeffs.clone()
            if i != 0 or coset_shift is not None:
                backend.distribute_powers(buf, gen)      # c_j * (g W^i)^j
            res.append(backend.batched_ntt(buf, 1, log_n, omega))   # res[t][k] = out[k*f + i]
    a = res[0] if fp == 1 else torch.cat(res)            # [fp][n]

    def interleave(part, log_len):
        """part: my cosets on a k range of 2^log_len values, [fp][2^log_len] -> my k block of it, [kb'][f]"""
        klen = (1 << log_len) // P
        # rank d owns k in [d*klen, (d+1)*klen) of the range, every coset: slab d = my cosets on that k range
        send = backend.pack(part, log_fp, log_len, log_p) if (fp > 1 and P > 1) else part
        if P == 1:
            recv = send
        else:
            recv = torch.empty_like(send)
            _all_to_all(recv, send, group)
        # slab s came from rank s and holds cosets i = s*fp + t: recv is [f][klen] in coset order -> [klen][f]
        return backend.transpose(recv, f, klen)

    if not paired:
        return interleave(a, log_n)
    assert n % (2 * P) == 0, "paired blocks: twice the world size must divide the polynomial size"
    halves = a.view(fp, 2, n // 2, a.shape[-1])
    return torch.cat([interleave(halves[:, h].contiguous().view(fp * (n // 2), a.shape[-1]), log_n - 1) for h in (0, 1)])


def lde_commit_by_cosets_distributed(ntt_backend, tree_backend, coeffs, log_n, factor, omega_big, rank, world,
                                     group=None, coset_shift=None, combiner=0):
    """LDE by cosets (one all-to-all) + distributed Merkle commit (one 32-byte all-gather).  combiner 1 (COSET2): the
    values come back in paired blocks and the tree is the COSET2 tree over all n*factor values."""
    lde_block = lde_by_cosets_distributed(ntt_backend, coeffs, log_n, factor, omega_big, rank, world, group,
                                          coset_shift, paired=bool(combiner))
    root, local_nodes, top = merkle_commit_distributed(tree_backend, lde_block, rank, world, group, combiner=combiner)
    return lde_block, root, local_nodes, top
