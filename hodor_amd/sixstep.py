"""sixstep.py — one NTT larger than a GPU, split across the ranks of a node (BASELINE config[4]).

6-step (Bailey) decomposition n = N1 * N2, the generalisation of the reference's own Cooley-Tukey
split in `parallel_fft` (/root/reference/src/fft/fft.rs:68-124: P sub-sequences, twiddle, sub-FFTs,
un-shuffle) to distributed memory, with every transpose realised as an all-to-all (RCCL over xGMI
when the process group is "nccl"; "gloo" in the CPU tests):

    x[n], n = n1*N2 + n2                rank r owns rows n1 in [r*N1/P, (r+1)*N1/P)   (natural blocks)
    1. all-to-all        -> rank q owns columns n2 in block q, all n1
    2. N2/P column NTTs of length N1 (omega^N2), times omega^(n2*k1)
    3. all-to-all        -> rank q owns k1 in block q, all n2
    4. N1/P row NTTs of length N2 (omega^N1)            -> X[k1 + N1*k2]
    5. all-to-all        -> natural blocks of X: rank q owns k in [q*n/P, (q+1)*n/P)

Natural order in, natural order out, bit-identical to a single-device transform (exact arithmetic).
The local arithmetic goes through a small backend interface so the same schedule runs on the HIP
kernels (HipBackend: hodor_fft_batch_dev / hodor_twiddle_mul_dev) and, in the CPU tests, on the
oracle.
"""
import torch
import torch.distributed as dist


class HipBackend:
    """Local steps on the MI355X through the C ABI (device tensors of shape (m, 4), int64)."""

    def __init__(self, ctx, stream=None):
        self.ctx, self.stream = ctx, stream

    def batched_ntt(self, buf, batch, log_len, omega):
        out = torch.empty_like(buf)
        self.ctx.fft_batch_dev(buf, out, log_len, batch, omega, stream=self.stream)
        return out

    def twiddle(self, buf, rows, cols, row0, omega, log_order, scale=None):
        self.ctx.twiddle_mul_dev(buf, rows, cols, row0, omega, log_order, scale=scale, stream=self.stream)
        return buf

    def distribute_powers(self, buf, g):
        self.ctx.distribute_powers_dev(buf, buf.shape[0], g, stream=self.stream)
        return buf

    def pow(self, a, e):
        return self.ctx.pow(a, e)

    def mul(self, a, b):
        return self.ctx.mul(a, b)

    def inverse(self, a):
        return self.ctx.inverse(a)

    def from_u64(self, v):
        return self.ctx.from_repr(v)


def _all_to_all_blocks(x, parts, group):
    """x: (parts, m, 4) — slab q goes to rank q; returns (parts, m, 4) with slab s received from rank s."""
    out = torch.empty_like(x)
    if parts == 1:
        out.copy_(x)
    else:
        dist.all_to_all_single(out, x.contiguous(), group=group)
    return out


def sixstep_ntt(backend, x_local, log_n, omega, rank, world, group=None, scale=None):
    """Distributed natural->natural NTT.  `x_local`: this rank's natural block, shape (n/world, 4).
    `omega`: Montgomery integer of a primitive n-th root (or its inverse).  `scale`: optional
    Montgomery scalar multiplied into every output (n^-1 for the inverse transform).
    Returns this rank's natural block of the transform."""
    n = 1 << log_n
    P = world
    log_n1 = log_n // 2
    log_n2 = log_n - log_n1
    N1, N2 = 1 << log_n1, 1 << log_n2
    assert N1 % P == 0 and N2 % P == 0, "world size must divide both factors"
    r1, c2 = N1 // P, N2 // P                      # my rows n1 / my columns n2

    # 1. (r1, N2) -> split columns into P slabs -> all-to-all -> (N1, c2) -> transpose to (c2, N1)
    a = x_local.view(r1, P, c2, 4).permute(1, 0, 2, 3).contiguous()        # (P, r1, c2)
    a = _all_to_all_blocks(a.view(P, r1 * c2, 4), P, group)                 # slab s = rows of rank s
    a = a.view(N1, c2, 4).permute(1, 0, 2).contiguous().view(c2 * N1, 4)   # (c2, N1): column-major

    # 2. column NTTs over n1 (length N1, root omega^N2), then * omega^(n2 * k1)
    w1 = backend.pow(omega, N2)
    a = backend.batched_ntt(a, c2, log_n1, w1)                              # Y[n2_local][k1]
    a = backend.twiddle(a, c2, N1, rank * c2, omega, log_n)

    # 3. (c2, N1) -> split k1 into P slabs -> all-to-all -> (N2, r1) -> transpose to (r1, N2)
    a = a.view(c2, P, r1, 4).permute(1, 0, 2, 3).contiguous()               # (P, c2, r1)
    a = _all_to_all_blocks(a.view(P, c2 * r1, 4), P, group)                 # slab s = n2 block of rank s
    a = a.view(N2, r1, 4).permute(1, 0, 2).contiguous().view(r1 * N2, 4)    # (r1 = my k1, N2)

    # 4. row NTTs over n2 (length N2, root omega^N1): Z[k1_local][k2] = X[k1 + N1*k2]
    w2 = backend.pow(omega, N1)
    a = backend.batched_ntt(a, r1, log_n2, w2)
    if scale is not None:
        a = backend.twiddle(a, r1, N2, 0, backend.from_u64(1), 0, scale=scale)

    # 5. natural blocks: rank q owns k2 in block q (all k1): (r1, N2) -> slabs over k2 -> all-to-all
    a = a.view(r1, P, c2, 4).permute(1, 0, 2, 3).contiguous()               # (P, r1, c2)
    a = _all_to_all_blocks(a.view(P, r1 * c2, 4), P, group)                 # slab s = k1 block of rank s
    a = a.view(N1, c2, 4).permute(1, 0, 2).contiguous().view(c2 * N1, 4)    # [k2_local][k1] = natural
    return a


def sixstep_intt(backend, x_local, log_n, omega, rank, world, group=None):
    """Inverse transform: omega^-1 and the n^-1 scale (Polynomial::ifft, src/polynomials/mod.rs:773-798)."""
    winv = backend.inverse(omega)
    ninv = backend.inverse(backend.from_u64(1 << log_n))
    return sixstep_ntt(backend, x_local, log_n, winv, rank, world, group, scale=ninv)
