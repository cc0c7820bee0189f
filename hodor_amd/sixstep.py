"""sixstep.py — one NTT larger than a GPU, split across the ranks of a node (BASELINE config[4]).

4-step / 6-step (Bailey) decomposition n = N1 * N2, the generalisation of the reference's own
Cooley-Tukey split in `parallel_fft` (/root/reference/src/fft/fft.rs:68-124: P sub-sequences, twiddle,
sub-FFTs, un-shuffle) to distributed memory.  All arithmetic AND all local data movement happen inside
the C ABI (include/hodor_gpu.h: hodor_sixstep_columns_dev / hodor_sixstep_rows_dev, with the transposes
fused into the transform kernels' addressing); this module only sequences the calls and performs the
exchange between them — ONE all-to-all of P contiguous slabs per transform (RCCL over xGMI when the
process group is "nccl"; "gloo" in the CPU tests).  A Rust caller binds the same entry points and does
the exchange with its own communicator.

Per-rank layouts (row-major, x[n1*N2 + n2], r1 = N1/P, c2 = N2/P, rank q):

    A  a[n1][j] = x[n1*N2 + q*c2 + j]      N1 x c2     column block q of the N1 x N2 input matrix
    B  b[i][k2] = X[(q*r1 + i) + N1*k2]    r1 x N2     row block q of the N1 x N2 output matrix

    sixstep_forward   A -> columns -> all-to-all -> rows -> B          (1 exchange)
    sixstep_inverse   B -> rows^-1 -> all-to-all -> columns^-1 -> A    (1 exchange, n^-1 folded in)
    sixstep_ntt / sixstep_intt   natural blocks in, natural blocks out: the same with one more exchange
                      and one copy (pack / transpose) on either side — 3 exchanges.  A prover that keeps
                      A/B between transforms (LDE -> pointwise work -> iNTT) never pays those.

The local steps go through a small backend interface so the same schedule runs on the HIP kernels
(HipBackend) and, in the CPU tests, on the oracle.

Round 5: with a library transport on the backend (`exchange` = hodor_amd.Exchange, RCCL behind the C ABI; `direct` =
hodor_amd.DirectExchange, the direct stores or the copy engine) the schedule below is NOT used: sixstep_forward /
_inverse (and their begin / end halves) are then one call each of the library's own schedule
(csrc/abi_dist.hip: hodor_dist_ntt_begin_dev / _end_dev), which is what a Rust process per GPU binds.  The Python
schedule remains for torch.distributed communicators — "nccl" process groups, and the "gloo" groups of the CPU tests
and of ranks that share one GPU — and as the readable twin of the C++ one.
"""
import torch
import torch.distributed as dist


class HipBackend:
    """Local steps on the MI355X through the C ABI (device tensors of shape (m, 4), int64)."""

    def __init__(self, ctx, stream=None, exchange=None, direct=None, direct_copy=False):
        # exchange: a hodor_amd.Exchange — the all-to-alls then run through the C ABI (hodor_sixstep_exchange_dev,
        # grouped ncclSend/ncclRecv on the library's communication stream) instead of torch.distributed
        # direct: a hodor_amd.DirectExchange — no all-to-all at all: the producing transform stores every slab straight
        # into the receive buffer of the rank it is for (hodor_sixstep_columns_direct_dev / _rows_direct_dev)
        # direct_copy: with `direct`, the copy-engine variant — the CHUNKED schedule writes its local send pieces as for an
        # all-to-all and the handle copies each into the peers' mapped buffers on its own stream (hodor_exchange_direct_copy_dev:
        # SDMA between devices, no CU, spread over whatever is enqueued next); consumer and flags are the direct transport's
        self.ctx, self.stream, self.exchange, self.direct, self.direct_copy = ctx, stream, exchange, direct, direct_copy

    # ---- 4-step building blocks
    def columns(self, src, log_n1, log_n2, log_p, rank, omega, inverse=False, log_chunks=0, chunk=0, out=None):
        """forward: column group `chunk` of A -> `out` (n/(P*K) elements); inverse: the K received chunk
        buffers -> A."""
        dst = out if out is not None else torch.empty_like(src)
        self.ctx.sixstep_columns_dev(src, dst, log_n1, log_n2, log_p, rank, omega, inverse, log_chunks, chunk,
                                     stream=self.stream)
        return dst

    def rows(self, src, log_n1, log_n2, log_p, rank, omega, inverse=False, log_chunks=0, chunk=0, out=None):
        """forward: the K received chunk buffers -> B; inverse: row group `chunk` of B -> `out`."""
        dst = out if out is not None else torch.empty_like(src)
        self.ctx.sixstep_rows_dev(src, dst, log_n1, log_n2, log_p, rank, omega, inverse, log_chunks, chunk,
                                  stream=self.stream)
        return dst

    def pack(self, src, log_rows, log_cols, log_p):
        if log_p == 0:
            return src
        dst = torch.empty_like(src)
        self.ctx.sixstep_pack_dev(src, dst, log_rows, log_cols, log_p, stream=self.stream)
        return dst

    def transpose(self, src, rows, cols):
        dst = torch.empty_like(src)
        self.ctx.transpose_dev(src, dst, rows, cols, stream=self.stream)
        return dst

    def scale(self, buf, s):
        self.ctx.poly_unary_dev(buf, buf.shape[0], "scale", c=s, stream=self.stream)
        return buf

    # ---- used by hodor_amd/distributed.py (LDE dealt by cosets)
    def batched_ntt(self, buf, batch, log_len, omega):
        out = torch.empty_like(buf)
        self.ctx.fft_batch_dev(buf, out, log_len, batch, omega, stream=self.stream)
        return out

    def distribute_powers(self, buf, g):
        self.ctx.distribute_powers_dev(buf, buf.shape[0], g, stream=self.stream)
        return buf

    def coset_ntt(self, coeffs, log_len, gen, omega):
        """values of sum_j coeffs[j] (gen x)^j on the size-2^log_len domain: the coset scale runs inside the first
        pass of the transform (coset_fft_for_generator, src/polynomials/mod.rs:633-638); gen None = plain fft."""
        out = torch.empty_like(coeffs)
        if gen is None:
            self.ctx.poly_fft_dev(coeffs, out, log_len, stream=self.stream)
        else:
            self.ctx.poly_coset_fft_for_generator_dev(coeffs, out, log_len, gen, stream=self.stream)
        return out

    def pow(self, a, e):
        return self.ctx.pow(a, e)

    def mul(self, a, b):
        return self.ctx.mul(a, b)

    def inverse(self, a):
        return self.ctx.inverse(a)

    def from_u64(self, v):
        return self.ctx.from_repr(v)


# Testing aid: issue the collectives even when world == 1 (a one-rank all-to-all is a valid RCCL call), so
# that the asynchronous-exchange / stream-ordering path can be exercised on a single-GPU box.
FORCE_COLLECTIVES = False


def split_logs(log_n):
    """n = N1 * N2 with N1 <= N2 (the choice every function of this module makes).  From 2^19 to 2^27 points
    N1 = 2^9: the column transforms are then ONE pass of the transform kernel (a 512-point sub-transform per
    workgroup tile) and the rows two, three passes over the data in all like the single-device plan, where the
    balanced split costs four (-4 ... -6 % local arithmetic at 2, 4, 8 ranks, profiles/r02/sixstep_rank_shape.txt).
    Outside that range the split is balanced."""
    log_n1 = 9 if 19 <= log_n <= 27 else log_n // 2
    return log_n1, log_n - log_n1


class _Done:
    def wait(self):
        return True


def _all_to_all(out, inp, group=None, async_op=False):
    """dist.all_to_all_single; device tensors on a "gloo" group (which has no device all-to-all) are staged
    through the host — a testing aid: two processes on ONE GPU can then run the real schedule end to end."""
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        hin = inp.cpu()                              # waits for the current stream's kernels
        hout = torch.empty_like(hin)
        dist.all_to_all_single(hout, hin, group=group)
        out.copy_(hout)
        return _Done()
    work = dist.all_to_all_single(out, inp, group=group, async_op=async_op)
    return work if async_op else _Done()


def all_to_all_slabs(x, world, group=None):
    """x: (m, 4) seen as `world` equal contiguous slabs; slab t goes to rank t.  Returns (m, 4) whose slab
    s came from rank s.  Bytes on the wire per rank: (world - 1) / world * m * 32."""
    if world == 1 and not FORCE_COLLECTIVES:
        return x
    out = torch.empty_like(x)
    _all_to_all(out, x, group)
    return out


def _log_p(world):
    log_p = world.bit_length() - 1
    assert 1 << log_p == world, "power-of-two world size"
    return log_p


def _exchange_begin(produce, m, world, group, log_chunks):
    """Runs produce(k, send_chunk_k) for k = 0 .. K-1 and puts each chunk on the wire as soon as it has been
    enqueued: the all-to-all of chunk k (asynchronous, on the communicator's own stream, ordered after the
    kernels that wrote the chunk) overlaps the arithmetic of chunk k+1 — and whatever the caller enqueues next.
    Returns (receive buffer, pending works); the receive buffer holds the chunk buffers back to back, the layout
    the consuming ABI call gathers from, once _exchange_end has been called."""
    K = 1 << log_chunks
    assert m % (K * world) == 0, "too many chunks for this transform"
    like = produce(None, None)                       # dtype/device probe: a tensor of the caller's kind
    send = torch.empty((m, 4), dtype=like.dtype, device=like.device)
    collective = world > 1 or FORCE_COLLECTIVES
    recv = torch.empty_like(send) if collective else send
    works = []
    step = m // K
    for k in range(K):
        produce(k, send[k * step:(k + 1) * step])
        if collective:
            works.append(_all_to_all(recv[k * step:(k + 1) * step], send[k * step:(k + 1) * step], group, async_op=True))
    return recv, works


def _exchange_end(works):
    for w in works:
        w.wait()                                     # the current stream waits for the exchange


def _library_transport(backend):
    """The exchange handle whose schedule lives in the library, or None (torch.distributed does the exchange)."""
    d = getattr(backend, "direct", None)
    if d is not None:
        d.set_transport(2 if getattr(backend, "direct_copy", False) else 1, FORCE_COLLECTIVES)
        return d
    x = getattr(backend, "exchange", None)
    if x is not None and hasattr(x, "dist_begin"):
        x.set_transport(0, FORCE_COLLECTIVES)
        return x
    return None


def sixstep_forward_begin(backend, a, log_n, omega, rank, world, group=None, log_chunks=0):
    """First half of sixstep_forward: the column transforms, chunk by chunk, each chunk's all-to-all started
    behind it.  The caller may enqueue unrelated work (the other half of another transform) before
    sixstep_forward_end, which waits for the exchange and runs the row transforms."""
    log_n1, log_n2 = split_logs(log_n)
    log_p = _log_p(world)
    x = _library_transport(backend)
    if x is not None:      # the library's own schedule: producer + exchange enqueued, `a` kept alive by the handle
        return {"lib": x, "op": x.dist_begin(a, log_n, omega, False, log_chunks, stream=backend.stream), "like": a}

    def produce(k, out):
        if k is None:
            return a
        backend.columns(a, log_n1, log_n2, log_p, rank, omega, False, log_chunks, k, out=out)

    recv, works = _exchange_begin(produce, a.shape[0], world, group, log_chunks)
    return {"recv": recv, "works": works, "args": (log_n1, log_n2, log_p, rank, omega, log_chunks)}


def sixstep_forward_end(backend, h):
    if "lib" in h:
        return h["lib"].dist_end(h["op"], torch.empty_like(h["like"]))
    log_n1, log_n2, log_p, rank, omega, log_chunks = h["args"]
    _exchange_end(h["works"])
    return backend.rows(h["recv"], log_n1, log_n2, log_p, rank, omega, False, log_chunks, 0)


def sixstep_forward(backend, a, log_n, omega, rank, world, group=None, log_chunks=0):
    """Layout A -> layout B, one exchange (cut into 2^log_chunks overlapped pieces).  `omega`: Montgomery
    integer of a primitive n-th root."""
    return sixstep_forward_end(backend, sixstep_forward_begin(backend, a, log_n, omega, rank, world, group, log_chunks))


def sixstep_inverse_begin(backend, b, log_n, omega, rank, world, group=None, log_chunks=0):
    """First half of sixstep_inverse: the inverse row transforms chunk by chunk with their all-to-alls."""
    log_n1, log_n2 = split_logs(log_n)
    log_p = _log_p(world)
    x = _library_transport(backend)
    if x is not None:
        return {"lib": x, "op": x.dist_begin(b, log_n, omega, True, log_chunks, stream=backend.stream), "like": b}

    def produce(k, out):
        if k is None:
            return b
        backend.rows(b, log_n1, log_n2, log_p, rank, omega, True, log_chunks, k, out=out)

    recv, works = _exchange_begin(produce, b.shape[0], world, group, log_chunks)
    return {"recv": recv, "works": works, "args": (log_n1, log_n2, log_p, rank, omega, log_chunks)}


def sixstep_inverse_end(backend, h):
    if "lib" in h:
        return h["lib"].dist_end(h["op"], torch.empty_like(h["like"]))
    log_n1, log_n2, log_p, rank, omega, log_chunks = h["args"]
    _exchange_end(h["works"])
    return backend.columns(h["recv"], log_n1, log_n2, log_p, rank, omega, True, log_chunks, 0)


def sixstep_inverse(backend, b, log_n, omega, rank, world, group=None, log_chunks=0):
    """Layout B -> layout A, one exchange; the inverse of sixstep_forward (same `omega`; n^-1 folded in)."""
    return sixstep_inverse_end(backend, sixstep_inverse_begin(backend, b, log_n, omega, rank, world, group, log_chunks))


def natural_to_a(backend, x_local, log_n, rank, world, group=None):
    """This rank's natural block (r1 rows of the N1 x N2 matrix) -> layout A: pack + one exchange."""
    log_n1, log_n2 = split_logs(log_n)
    log_p = _log_p(world)
    return all_to_all_slabs(backend.pack(x_local, log_n1 - log_p, log_n2, log_p), world, group)


def b_to_natural(backend, b, log_n, rank, world, group=None):
    """Layout B -> this rank's natural block of the output: pack + one exchange + a local transpose."""
    log_n1, log_n2 = split_logs(log_n)
    log_p = _log_p(world)
    y = all_to_all_slabs(backend.pack(b, log_n1 - log_p, log_n2, log_p), world, group)
    return backend.transpose(y, 1 << log_n1, 1 << (log_n2 - log_p))      # [k1][k2_local] -> [k2_local][k1]


def sixstep_ntt(backend, x_local, log_n, omega, rank, world, group=None, scale=None):
    """Distributed natural->natural NTT (3 exchanges).  `x_local`: this rank's natural block, shape
    (n/world, 4).  `scale`: optional Montgomery scalar multiplied into every output."""
    x = _library_transport(backend)
    if x is not None and scale is None:
        return x.dist_natural(x_local, torch.empty_like(x_local), log_n, omega, False, stream=backend.stream)
    a = natural_to_a(backend, x_local, log_n, rank, world, group)
    b = sixstep_forward(backend, a, log_n, omega, rank, world, group)
    if scale is not None:
        b = backend.scale(b, scale)
    return b_to_natural(backend, b, log_n, rank, world, group)


def sixstep_intt(backend, x_local, log_n, omega, rank, world, group=None):
    """Inverse natural->natural transform: omega^-1 and the n^-1 scale (Polynomial::ifft,
    src/polynomials/mod.rs:773-798)."""
    x = _library_transport(backend)
    if x is not None:
        return x.dist_natural(x_local, torch.empty_like(x_local), log_n, omega, True, stream=backend.stream)
    winv = backend.inverse(omega)
    ninv = backend.inverse(backend.from_u64(1 << log_n))
    return sixstep_ntt(backend, x_local, log_n, winv, rank, world, group, scale=ninv)
