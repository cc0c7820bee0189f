// abi_sixstep.hip — building blocks of ONE transform split over the P GPUs of a node (SURVEY.md §8(e),
// BASELINE config[4]): the 4-step / 6-step (Bailey) decomposition n = N1 * N2, the distributed-memory
// form of the reference's own Cooley-Tukey split in parallel_fft (/root/reference/src/fft/fft.rs:68-124:
// shuffle into P sub-sequences, twiddle, sub-FFTs, un-shuffle).
//
// The library does the local arithmetic with the layout changes FUSED into the pass kernel's
// addressing (k_ntt_pass<1>: column mode, 2D twiddle, split addressing); the caller owns the
// communicator and performs the exchange between the two calls — one all-to-all of P equal,
// contiguous slabs (RCCL over xGMI: torch.distributed all_to_all_single in hodor_amd/sixstep.py,
// ncclSend/ncclRecv pairs from Rust).  x[n1*N2 + n2], r1 = N1/P, c2 = N2/P, rank q:
//
//   layout A (column blocks)   a[n1][j]  = x[n1*N2 + q*c2 + j]          N1 x c2, row-major
//   layout B (row blocks)      b[i][k2]  = X[(q*r1 + i) + N1*k2]        r1 x N2, row-major
//
//   forward  A -> B:  columns (N1-point transforms down the columns, times w^(k1*n2); output rows k1,
//                     so the slab for rank t — rows t*r1 .. — is contiguous)
//                     -> all-to-all -> rows (N2-point transforms; the input element n2 = s*c2 + j of row
//                     i sits in the slab received from rank s: split addressing, no unpack copy)
//   inverse  B -> A:  rows^-1 (output written slab by slab: split addressing, no pack copy)
//                     -> all-to-all -> columns^-1 (times w^-(k1*n2) on load, n^-1 folded in)
//
// One exchange per transform, optionally cut into K chunks so that chunk k's all-to-all overlaps the
// arithmetic of chunk k+1 (forward: the columns are transformed K groups at a time; inverse: the rows).
// Natural block order on either side costs one more exchange each plus
// hodor_sixstep_pack_dev / hodor_transpose_dev (a caller that keeps A/B between transforms — NTT,
// pointwise work, iNTT — never pays them).
#include "ctx.hpp"

#ifdef HODOR_BOUNDS
// abi_exchange.hip: the receive buffers of a slot as the host knows them; returns the size of the library's own ones (0: the caller's)
extern "C" uint64_t hodor_exchange_direct_host_table(hodor_exchange *x, uint32_t slot, uint64_t out[8]);
#endif

namespace hodor {

// dst[c][r] = src[r][c] for 32-byte elements; 16 x 16 tiles through LDS (512-byte runs both ways)
__global__ void __launch_bounds__(256)
k_transpose(const uint4 *src, uint4 *dst, uint64_t rows, uint64_t cols, uint32_t tiles_c BXPARAM)
{
    __shared__ uint4 lo[16][17], hi[16][17];
    const uint32_t tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const uint64_t by = blockIdx.x / tiles_c, bx = blockIdx.x % tiles_c;
    uint64_t r = by * 16 + ty, c = bx * 16 + tx;
    if (r < rows && c < cols) {
        const uint4 *s = BAT(1, src, 2 * (r * cols + c), 2);
        lo[ty][tx] = s[0];
        hi[ty][tx] = s[1];
    }
    __syncthreads();
    r = by * 16 + tx;
    c = bx * 16 + ty;
    if (r < rows && c < cols) {
        uint4 *d = BATS(2, dst, 2 * (c * rows + r), 2);
        d[0] = lo[tx][ty];
        d[1] = hi[tx][ty];
    }
}

// dst[(t*rows + i)*c2 + j] = src[i*(P*c2) + t*c2 + j]: a rows x (P*c2) block cut into the P slabs of an
// all-to-all (runs of c2 contiguous elements)
__global__ void __launch_bounds__(256)
k_pack_slabs(const uint4 *src, uint4 *dst, uint32_t log_rows, uint32_t log_c2, uint32_t log_p BXPARAM)
{
    const uint64_t total = 1ull << (log_rows + log_c2 + log_p);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        uint64_t j = e & ((1ull << log_c2) - 1);
        uint64_t t = (e >> log_c2) & ((1ull << log_p) - 1);
        uint64_t i = e >> (log_c2 + log_p);
        uint64_t d = (((t << log_rows) + i) << log_c2) + j;
        const uint4 *s = BAT(1, src, 2 * e, 2);
        uint4 a = s[0], b = s[1];
        *BATS(2, dst, 2 * d, 2) = a;
        dst[2 * d + 1] = b;
    }
}

}  // namespace hodor

static int sixstep_check(hodor_ctx *ctx, uint32_t log_n1, uint32_t log_n2, uint32_t log_p, uint32_t rank)
{
    if (log_p > log_n1 || log_p > log_n2 || rank >= (1u << log_p) || log_n1 + log_n2 > ctx->F.s ||
        log_n1 + log_n2 > 40) {
        set_err(ctx, "sixstep: need P | N1, P | N2, rank < P, log2(N1 N2) <= the field's 2-adicity");
        return HODOR_ERR_SIZE;
    }
    return HODOR_OK;
}

// An axis that arrives (or leaves) cut into K chunks x P slabs: element x = s*c + k*cw + j (c = K*cw per
// slab) of batch member i lives at  k*(P*b*cw) + s*(b*cw) + i*cw + j,  b = batch members per chunk buffer.
static SplitAddr chunked_slab_axis(uint32_t log_b, uint32_t log_c, uint32_t log_p, uint32_t log_k)
{
    SplitAddr S = {};
    if (log_p == 0 && log_k == 0) return S;     // one slab, one chunk: plain rows
    const uint32_t log_cw = log_c - log_k;
    S.on = 1;
    S.lo_log = log_cw;
    S.mid_mask = (1u << log_k) - 1;
    S.stride_mid = 1ull << (log_p + log_b + log_cw);
    S.hi_log = log_c;
    S.stride_hi = 1ull << (log_b + log_cw);
    S.batch_stride = 1ull << log_cw;
    return S;
}

static int chunk_check(hodor_ctx *ctx, uint32_t log_avail, uint32_t log_chunks, uint32_t chunk)
{
    if (log_chunks > log_avail || chunk >= (1u << log_chunks)) {
        set_err(ctx, "sixstep: chunk index / count out of range");
        return HODOR_ERR_SIZE;
    }
    return HODOR_OK;
}

extern "C" int hodor_sixstep_columns_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_fr *dst,
                                         uint32_t log_n1, uint32_t log_n2, uint32_t log_p, uint32_t rank,
                                         const hodor_fr *omega, int inverse, uint32_t log_chunks, uint32_t chunk)
{
    NEED_DEVICE();
    if (!src || !dst || !omega || src == dst) return HODOR_ERR_INVALID;
    int rc = sixstep_check(ctx, log_n1, log_n2, log_p, rank);
    if (rc) return rc;
    const uint32_t log_r1 = log_n1 - log_p, log_c2 = log_n2 - log_p, log_n = log_n1 + log_n2;
    // forward: the chunks cut the columns (this call transforms chunk `chunk`); inverse: they cut the rows
    // of the buffers this call gathers from (all of them at once)
    if ((rc = chunk_check(ctx, inverse ? log_r1 : log_c2, log_chunks, inverse ? 0 : chunk))) return rc;
    HFr w = to_h(omega);
    if (inverse && !ctx->F.inverse(w, &w)) return HODOR_ERR_INVALID;
    HFr w1 = ctx->F.pow(w, 1ull << log_n2);                 // primitive N1-th root
    HFr ninv;
    if (inverse) ctx->F.inverse(ctx->F.from_u64(1ull << log_n), &ninv);
    NttLayout L;
    L.col_mode = true;
    L.tw2d_root = &w;
    L.tw2d_log_order = log_n;
    L.tw2d_on_load = inverse != 0;
    if (!inverse) {
        const uint32_t log_cw = log_c2 - log_chunks;
        L.log_width = log_cw;                               // compact [N1][cw] out: its P slabs are contiguous
        L.src_log_width = log_c2;
        L.src_col_off = (uint64_t)chunk << log_cw;
        L.col0 = ((uint64_t)rank << log_c2) + ((uint64_t)chunk << log_cw);
    } else {
        L.log_width = log_c2;
        L.col0 = (uint64_t)rank << log_c2;
        if (log_chunks) {   // row k1 = s*r1 + b*rb + i sits in chunk buffer b, slab s, row i: [K][P][rb][c2]
            const uint32_t log_rb = log_r1 - log_chunks;
            SplitAddr S = {};
            S.on = 1;
            S.lo_log = log_rb;
            S.mid_mask = (1u << log_chunks) - 1;
            S.stride_mid = 1ull << (log_p + log_rb);
            S.hi_log = log_r1;
            S.stride_hi = 1ull << log_rb;
            S.batch_stride = 0;
            L.src_split = S;
        }
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    return ntt_exec(ctx, pick_stream(ctx, stream), (const uint4 *)src, (uint4 *)dst, log_n1, w1, 1ull << log_n1,
                    inverse ? &ninv : nullptr, nullptr, nullptr, 1u << L.log_width, &L);
}

extern "C" int hodor_sixstep_rows_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_fr *dst,
                                      uint32_t log_n1, uint32_t log_n2, uint32_t log_p, uint32_t rank,
                                      const hodor_fr *omega, int inverse, uint32_t log_chunks, uint32_t chunk)
{
    NEED_DEVICE();
    if (!src || !dst || !omega || src == dst) return HODOR_ERR_INVALID;
    int rc = sixstep_check(ctx, log_n1, log_n2, log_p, rank);
    if (rc) return rc;
    const uint32_t log_r1 = log_n1 - log_p, log_c2 = log_n2 - log_p;
    if ((rc = chunk_check(ctx, inverse ? log_r1 : log_c2, log_chunks, inverse ? chunk : 0))) return rc;
    HFr w = to_h(omega);
    if (inverse && !ctx->F.inverse(w, &w)) return HODOR_ERR_INVALID;
    HFr w2 = ctx->F.pow(w, 1ull << log_n1);                 // primitive N2-th root
    NttLayout L;
    uint32_t log_rows = log_r1;
    if (!inverse) {
        L.src_split = chunked_slab_axis(log_r1, log_c2, log_p, log_chunks);   // gathers all K chunk buffers
    } else {
        log_rows = log_r1 - log_chunks;                     // this call: rows chunk*rb .. of B
        src += ((size_t)chunk << log_rows) << log_n2;
        L.dst_split = chunked_slab_axis(log_rows, log_c2, log_p, 0);
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    return ntt_exec(ctx, pick_stream(ctx, stream), (const uint4 *)src, (uint4 *)dst, log_n2, w2, 1ull << log_n2,
                    nullptr, nullptr, nullptr, 1u << log_rows, &L);
}

// ---- the producers of the direct exchange (abi_exchange.hip): the same transforms with their last pass storing slab t
// into rank t's receive buffer (slot `slot` of the handle) instead of a local send buffer; forward = the column
// transforms, inverse = the inverse row transforms.  The consumers are the plain calls above on the local buffer.
extern "C" int hodor_exchange_direct_table(hodor_exchange *x, uint32_t slot, const uint64_t **tab, uint32_t *n_ranks,
                                           uint32_t *rank);

extern "C" int hodor_sixstep_columns_direct_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_exchange *x,
                                                uint32_t slot, uint32_t log_n1, uint32_t log_n2, const hodor_fr *omega,
                                                uint32_t log_chunks, uint32_t chunk)
{
    NEED_DEVICE();
    if (!src || !omega || !x) return HODOR_ERR_INVALID;
    const uint64_t *tab = nullptr;
    uint32_t P = 0, rank = 0;
    if (hodor_exchange_direct_table(x, slot, &tab, &P, &rank)) { set_err(ctx, "sixstep (direct): slot not set up"); return HODOR_ERR_INVALID; }
    const uint32_t log_p = log2u(P);
    int rc = sixstep_check(ctx, log_n1, log_n2, log_p, rank);
    if (rc) return rc;
    const uint32_t log_r1 = log_n1 - log_p, log_c2 = log_n2 - log_p, log_n = log_n1 + log_n2;
    if ((rc = chunk_check(ctx, log_c2, log_chunks, chunk))) return rc;
    HFr w = to_h(omega);
    HFr w1 = ctx->F.pow(w, 1ull << log_n2);
    const uint32_t log_cw = log_c2 - log_chunks;
    NttLayout L;
    L.col_mode = true;
    L.tw2d_root = &w;
    L.tw2d_log_order = log_n;
    L.tw2d_on_load = false;
    L.log_width = log_cw;
    L.src_log_width = log_c2;
    L.src_col_off = (uint64_t)chunk << log_cw;
    L.col0 = ((uint64_t)rank << log_c2) + ((uint64_t)chunk << log_cw);
    L.peer_tab = tab;
    L.peer_log = log_r1;                                   // output row k1 goes to rank k1 >> log_r1 ...
    L.peer_self = rank;                                    // ... as row rank*r1 + (k1 mod r1) of its [P][r1][cw] chunk buffer
    L.peer_off = (uint64_t)chunk << (log_n1 + log_cw);     // chunk buffers back to back: n_local / K elements each
#ifdef HODOR_BOUNDS
    L.bx_peers = P;
    L.bx_peer_bytes = hodor_exchange_direct_host_table(x, slot, L.bx_peer_host);
    if (!L.bx_peer_bytes) L.bx_peer_bytes = 32ull << (log_n - log_p);   // the caller's own receive buffers: n / P elements by contract
#endif
    std::lock_guard<std::mutex> lk(ctx->mu);
    return ntt_exec(ctx, pick_stream(ctx, stream), (const uint4 *)src, nullptr, log_n1, w1, 1ull << log_n1, nullptr, nullptr,
                    nullptr, 1u << L.log_width, &L);
}

extern "C" int hodor_sixstep_rows_direct_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_exchange *x,
                                             uint32_t slot, uint32_t log_n1, uint32_t log_n2, const hodor_fr *omega,
                                             uint32_t log_chunks, uint32_t chunk)
{
    NEED_DEVICE();
    if (!src || !omega || !x) return HODOR_ERR_INVALID;
    const uint64_t *tab = nullptr;
    uint32_t P = 0, rank = 0;
    if (hodor_exchange_direct_table(x, slot, &tab, &P, &rank)) { set_err(ctx, "sixstep (direct): slot not set up"); return HODOR_ERR_INVALID; }
    const uint32_t log_p = log2u(P);
    int rc = sixstep_check(ctx, log_n1, log_n2, log_p, rank);
    if (rc) return rc;
    const uint32_t log_r1 = log_n1 - log_p, log_c2 = log_n2 - log_p;
    if ((rc = chunk_check(ctx, log_r1, log_chunks, chunk))) return rc;
    HFr w = to_h(omega);
    if (!ctx->F.inverse(w, &w)) return HODOR_ERR_INVALID;
    HFr w2 = ctx->F.pow(w, 1ull << log_n1);
    const uint32_t log_rows = log_r1 - log_chunks;         // this call: rows chunk*rb .. of B
    src += ((size_t)chunk << log_rows) << log_n2;
    NttLayout L;
    SplitAddr S = {};                                      // element n2 = s*c2 + j of row i -> slab s, [rb][c2]
    S.on = 1;
    S.lo_log = log_c2;
    S.mid_mask = 0;
    S.stride_mid = 0;
    S.hi_log = log_c2;
    S.stride_hi = 1ull << (log_rows + log_c2);
    S.batch_stride = 1ull << log_c2;
    L.dst_split = S;
    L.peer_tab = tab;
    L.peer_self = rank;
    L.peer_off = (uint64_t)chunk << (log_p + log_rows + log_c2);   // [K][P][rb][c2]
#ifdef HODOR_BOUNDS
    L.bx_peers = P;
    L.bx_peer_bytes = hodor_exchange_direct_host_table(x, slot, L.bx_peer_host);
    if (!L.bx_peer_bytes) L.bx_peer_bytes = 32ull << (log_n1 + log_n2 - log_p);
#endif
    std::lock_guard<std::mutex> lk(ctx->mu);
    return ntt_exec(ctx, pick_stream(ctx, stream), (const uint4 *)src, nullptr, log_n2, w2, 1ull << log_n2, nullptr, nullptr,
                    nullptr, 1u << log_rows, &L);
}

extern "C" int hodor_sixstep_pack_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_fr *dst,
                                      uint32_t log_rows, uint32_t log_cols, uint32_t log_p)
{
    NEED_DEVICE();
    if (!src || !dst || src == dst) return HODOR_ERR_INVALID;
    if (log_p > log_cols || log_rows + log_cols > 40) return HODOR_ERR_SIZE;
    uint64_t total = 1ull << (log_rows + log_cols);
    uint64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    BX_BEGIN(bx, KID_SIXSTEP_PACK);
    BX_ADD(bx, src, total * 32);
    BX_ADD(bx, dst, total * 32);
    hipLaunchKernelGGL(k_pack_slabs, dim3((unsigned)blocks), dim3(256), 0, pick_stream(ctx, stream),
                       (const uint4 *)src, (uint4 *)dst, log_rows, log_cols - log_p, log_p BXARG(bx));
    HIPCHK(hipGetLastError());
    return HODOR_OK;
}

extern "C" int hodor_transpose_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_fr *dst, size_t rows,
                                   size_t cols)
{
    NEED_DEVICE();
    if (!src || !dst || src == dst) return HODOR_ERR_INVALID;
    if (rows == 0 || cols == 0) return HODOR_OK;
    const uint64_t tiles_c = (cols + 15) / 16, tiles_r = (rows + 15) / 16;
    if (tiles_c * tiles_r > 0x7fffffffull || tiles_c > 0xffffffffull) { set_err(ctx, "transpose: too many tiles"); return HODOR_ERR_SIZE; }
    BX_BEGIN(bx, KID_TRANSPOSE);
    BX_ADD(bx, src, (uint64_t)rows * cols * 32);
    BX_ADD(bx, dst, (uint64_t)rows * cols * 32);
    hipLaunchKernelGGL(k_transpose, dim3((unsigned)(tiles_c * tiles_r)), dim3(256), 0, pick_stream(ctx, stream),
                       (const uint4 *)src, (uint4 *)dst, (uint64_t)rows, (uint64_t)cols, (uint32_t)tiles_c BXARG(bx));
    HIPCHK(hipGetLastError());
    return HODOR_OK;
}
