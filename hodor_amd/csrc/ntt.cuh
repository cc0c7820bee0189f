// ntt.cuh — shared declarations for the NTT pass kernels (K2/K3/K4/K5).
#pragma once
#include "fr9.cuh"

namespace hodor {

// Power tables for one (base, log_n) pair, in device memory:
//   lo[j] = base^j                          j < 2^lo_bits
//   hi[j] = scale * base^(j << lo_bits)     j < 2^(log_n - lo_bits)     (scale folds n^-1 for the iNTT)
// so base^e = hi[e >> lo_bits] * lo[e & mask], one extra product (or none when the low part of e is known
// to be zero).  Entry format by consumer (get_pow_table's fmt): 112-byte W3 constants for k_ntt_pass, which
// applies the two halves as two successive data x constant products (fr9w3.cuh); 48-byte R'-form 9 x 29-bit
// values where the two halves are multiplied together first (FRI fold, distribute_powers, evaluate_at);
// 32-byte R-form values (twiddle_mul).  The reference recomputes twiddles by running products
// (src/fft/fft.rs:58) or tabulates all n of them (src/precomputations/mod.rs:14-66); a two-level table
// keeps the working set L2-sized instead of streaming n entries from HBM.
struct TwoLevel {
    const uint4 *lo;
    const uint4 *hi;
    uint32_t lo_bits;
#ifdef HODOR_BOUNDS
    uint64_t lo_bytes, hi_bytes;   // sizes of the two tables, for the launchers' extent declarations (bounds.cuh)
#endif
};

// 2^87, 2^174, 2^261 mod p as plain integers: turn an R-form power into its W3 entry (fr9w3.cuh)
struct W3Consts {
    Fr k[3];
};

// 2^(29 (c + 1)) mod p, c = 0 .. 8, as plain integers: the W9 entry of an R-form power (fr9w3.cuh)
struct W9Consts {
    Fr k[9];
};
constexpr int W9_WORDS = 108;   // a W9 entry: 9 columns of 9 limbs, each column padded to 12 words

// Split addressing of one transform axis (6-step building blocks, abi_sixstep.hip): element x of batch
// member b lives at
//     (x >> hi_log) * stride_hi + ((x >> lo_log) & mid_mask) * stride_mid + b * batch_stride + (x & lo_mask)
// i.e. the axis arrives (or leaves) cut into P slabs of an all-to-all, each slab optionally cut into
// `chunks` pieces that were exchanged separately.  on == 0: plain arrays.
struct SplitAddr {
    uint32_t on, lo_log, hi_log, mid_mask;
    uint64_t stride_mid, stride_hi, batch_stride;
};

struct PassArgs {
    const uint4 *src;        // n elements (only the first nnz are read; the rest are implicit zeros)
    uint4 *dst;              // n elements
    const uint4 *rtw;        // omega_R^e, e < R/2  (R = radix of this pass), W3 entries of 7 x 16 B
    TwoLevel tw;             // inter-pass twiddles (powers of the size-n omega; hi possibly scaled)
    TwoLevel pre;            // optional input scaling x[i] *= g^i  (coset shift), lo == nullptr if unused
    TwoLevel post;           // optional output scaling X[k] *= h^k, lo == nullptr if unused
    uint64_t nnz;            // number of leading non-zero inputs (== n unless LDE zero padding)
    uint32_t log_n;          // transform size
    uint32_t log_r;          // radix of this pass (R = 2^log_r points per sub-transform)
    uint32_t log_c;          // tile columns (C consecutive sub-transforms per workgroup)
    uint32_t log_l;          // product of the radices of the previous passes (L = 2^log_l)
    uint32_t apply_tw;       // 0: no inter-pass twiddle (first pass)  1: lo*hi  2: hi only
    uint32_t tw_always;      // 1: multiply by the twiddle even when its exponent is 0 (hi carries the iNTT scale)
    uint32_t batch;          // number of independent size-n transforms (grid.y); dst arrays are n elements apart
    uint64_t src_batch_stride;   // distance between the batch's source arrays, in elements
    const uint32_t *rtw9;    // omega_R^(e * R/32), e < 16, as W9 entries (fr9w3.cuh) for the wave-uniform steps; may be null
    uint32_t w9_limit;       // set by the launcher: radix-4 steps with half-size m < 2^this take their twiddles from
                             //   rtw9 (0: none), items dealt so that a wave works on ONE twiddle set
    uint32_t w9_skip_one;    // 1: a wave whose twiddle index is 0 skips the products by one
    uint32_t tw_sub;         // set by the launcher: the LDS twiddle table holds every 2^tw_sub-th entry (see k_ntt_pass)
    uint32_t log_skip;       // first pass of a zero-padded transform: nnz == n >> log_skip (see k_ntt_pass)
    // ---- generalized layouts (k_ntt_pass<1> only; all zero for plain arrays)
    uint32_t col_mode;       // 1: the data is a 2D array [index][width] and the transform runs along `index` for
                             //    every column: the tile's C columns are C adjacent array columns (grid.y = width/C)
                             //    and a workgroup owns ONE sub-transform position j
    uint32_t log_width;      // col_mode: log2(width) of the destination (and of every intermediate) array
    uint32_t src_log_width;  // col_mode, first pass: log2(width) of the SOURCE array, and the array column of
    uint64_t src_col_off;    //   its tile column 0 (a chunk of the columns of a wider array is transformed)
    uint64_t dst_col_off;    // col_mode, last pass: likewise for a destination wider than the chunk (dst_log_width)
    uint32_t dst_log_width;
    uint64_t col0;           // col_mode: global index of array column 0, for the 2D twiddle w^(index * (col0 + col))
    TwoLevel tw2d;           // col_mode: that twiddle's table (lo == nullptr: none) ...
    uint32_t tw2d_on_load;   // ... applied to the inputs of the first pass (1) or the outputs of the last pass (0)
    SplitAddr src_split;     // first pass: where input element x of batch member b lives
    SplitAddr dst_split;     // last pass: where output element x of batch member b goes
    // ---- direct exchange (abi_exchange.hip, "direct" transport): the last pass stores every output slab straight into
    // the receive buffer of the rank it is for, so the all-to-all of the 4-step transform disappears into the store phase
    const uint64_t *peer_tab; // device array: peer_tab[t] = address of rank t's receive buffer as mapped in this process (null: off)
    uint64_t peer_off;       // element offset of this call's chunk buffer inside every receive buffer
    uint32_t peer_log;       // column mode: log2(rows per slab): output row o goes to rank o >> peer_log, as row
    uint32_t peer_self;      //   (peer_self << peer_log) + (o mod rows per slab); split mode: rank x >> hi_log, slab peer_self
    uint32_t dbg;            // read only by -DHODOR_ABLATE builds (bench/ablate.sh): 1 skip butterflies, 2 twiddles, 4 loads, 8 stores
#ifdef HODOR_BOUNDS
    // host side only (ntt_exec -> ntt_launch_pass): the extents of the buffers of this pass (bounds.cuh)
    uint64_t bx_src_bytes, bx_dst_bytes, bx_rtw_bytes, bx_rtw9_bytes, bx_peer_bytes;
    uint64_t bx_peer_host[8];    // the receive buffers peer_tab points at, as the host knows them
    uint32_t bx_peers;
#endif
};

// One FRI folding step (src/fri/fri_on_values.rs:77-100), shared by k_fri_fold (fri.hip) and the fused
// fold + leaf-hash phase of the latency-schedule Merkle kernel (merkle.hip).
struct FoldArgs {
    const uint4 *src;        // 2 * half values of the current round
    uint4 *dst;              // half values of the next round
    uint64_t half;
    const uint4 *lo;         // two-level table of w^-1 of the initial domain (R'-form) ...
    const uint4 *hi_beta;    // ... its `hi` half scaled by beta / 2 for this round (k_fri_round_table)
    uint32_t lo_bits;
    uint32_t log_stride;     // round index: element i pairs with exponent i << log_stride
#ifdef HODOR_BOUNDS
    uint64_t lo_bytes, hi_bytes;   // sizes of the two tables (host side, for the launchers' extent declarations)
#endif
};

// exact halving of a lazy value: add p when odd, shift right one bit across the 29-bit limbs
__device__ __forceinline__ Fr9 fr9_halve(Fr9 a, const Fr9Params &Q)
{
    fr9_normalize(a);
    uint32_t odd = a.v[0] & 1;
#pragma unroll
    for (int i = 0; i < 9; i++) a.v[i] += odd ? Q.p[i] : 0u;
    fr9_normalize(a);
    Fr9 r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = (a.v[i] >> 1) | ((a.v[i + 1] & 1) << 28);
    r.v[8] = a.v[8] >> 1;
    return r;
}

// next[i] = (f[i] + f[i+half]) / 2 + (f[i] - f[i+half]) * w^-(i << log_stride) * beta / 2, canonical
__device__ __forceinline__ Fr fri_fold_one(const FoldArgs &F, uint64_t i, const Fr9Params &Q BXPARAM)
{
    Fr9 a = fr9_unpack(fr_load(BAT(40, F.src, 2 * i, 2))), b = fr9_unpack(fr_load(BAT(41, F.src, 2 * (i + F.half), 2)));
    const uint64_t e = i << F.log_stride, lo_mask = (1ull << F.lo_bits) - 1;
    Fr9 tw = fr9_load48(BAT(42, F.hi_beta, 3 * (e >> F.lo_bits), 3));
    if (e & lo_mask) tw = fr9_mul(tw, fr9_load48(BAT(43, F.lo, 3 * (e & lo_mask), 3)), Q);
    Fr9 odd = fr9_mul(fr9_sub(a, b, Q), tw, Q);          // (a - b) * beta * w^-e / 2
    Fr9 even = fr9_halve(fr9_add(a, b), Q);              // (a + b) / 2
    return fr9_to_canonical(fr9_add(even, odd), Q);
}

// Arguments of the fused FRI tail (fri.hip, k_fri_tail): the rounds whose output is <= FRI_TAIL_THREADS
// values, run by one workgroup.
constexpr int FRI_TAIL_THREADS = 512;
constexpr int FRI_TAIL_MAX_ROUNDS = 10;
struct FriTailArgs {
    const uint4 *src;                       // values entering the first fused round (2 * half0 elements)
    uint4 *values[FRI_TAIL_MAX_ROUNDS];     // output of fused round k: half0 >> k elements
    uint4 *nodes[FRI_TAIL_MAX_ROUNDS];      // tree over values[k]
    uint4 *chal;                            // challenges: entry i enters round i; round i's tree writes entry i + 1
    uint4 *roots;                           // roots: round i's tree writes entry i + 1
    const uint4 *lo, *hi;                   // two-level table of w^-1 of the initial domain (R'-form)
    uint32_t lo_bits;
    uint32_t rounds;                        // number of fused rounds
    uint32_t first_round;                   // index i of the first fused round
    uint32_t half0;                         // outputs of the first fused round (power of two, 2 .. FRI_TAIL_THREADS)
    uint32_t shave;                         // 256 - CAPACITY
#ifdef HODOR_BOUNDS
    uint64_t lo_bytes, hi_bytes;            // sizes of the two tables (host side: extent declarations)
#endif
};

}  // namespace hodor
