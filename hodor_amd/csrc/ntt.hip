// ntt.hip — K2/K3/K4/K5: power tables + Stockham multi-pass NTT over a 256-bit prime field.
//
// What it replaces in the reference (paths relative to /root/reference):
//   best_fft / serial_fft / parallel_fft       src/fft/fft.rs:5-124        (natural -> natural DFT)
//   serial_fft_radix_4                         src/fft/radix4_fft/mod.rs:45-123
//   best_lde (zero-aware FFT)                  src/fft/lde.rs:4-193
//   distribute_powers fused as pre/post scale  src/fft/mod.rs:110-123
// All field arithmetic is exact, so any schedule that computes A[k] = sum_i a[i] w^(ik) mod p in
// natural order is bit-identical to the CPU path.  The schedule here is MI355X-first:
//
//   * The transform of size n = R_1 * R_2 * ... is done in a few passes; pass s computes n/R_s
//     independent R_s-point sub-transforms (Stockham autosort indexing -> natural order out, no
//     bit-reversal pass over HBM).  One workgroup owns a tile of C adjacent sub-transforms, so every
//     global read and write is a run of C*32 contiguous bytes.
//   * Inside the workgroup the R-point transform runs out of LDS (in-place DIT, two radix-2 stages
//     fused per barrier = one radix-4 step), twiddles of the sub-transform staged in LDS.
//   * Inside the kernel elements live in the carry-free 9 x 29-bit form of fr9.cuh (add/sub = 9
//     plain adds, lazily reduced); HBM always holds the reference's 8 x 32-bit Montgomery image.
//     LDS keeps limbs 0-3 / 4-7 as two 16-byte arrays plus a 4-byte array for limb 8, columns
//     XOR-swizzled by the row index.
//   * Butterfly twiddles sit in LDS as W3 constants (fr9w3.cuh: the twiddle pre-shifted three ways,
//     112 B per entry), so a butterfly product is 108 v_mad_u64_u32 + 3 v_mul_lo instead of 162 + 9.
//   * Inter-pass twiddles and coset powers come from two-level power tables of W3 constants
//     (base^e = hi[e >> b] * lo[e & mask], applied as two successive products) instead of an n-entry
//     table streamed from HBM.
//
// Value bounds inside a pass (p < 2^255, 2^261 > 64 p): loaded values are < 2^256 < 4 p (a packed
// value) or < 4 p (after a twiddle product); a W3 product of a normalized value is < 4 p; subtraction
// adds the 5 p offset, so a radix-4 step raises the bound by at most 10 p (two subtractions):
// 14 p after the twiddle-free first step (9 p after a leading radix-2 stage), <= 59 p after the
// <= 6 steps of a pass of radix <= 2^11.  Every subtrahend is a fresh product (< 4 p) except in the
// twiddle-free first step, handled explicitly; every product takes a normalized operand.
#include <cstdlib>
#include <type_traits>

#include "knobs.hpp"
#include "ntt.cuh"
#include "fr9w3.cuh"

namespace hodor {

// ---------------------------------------------------------------------------------------------
// K2: out[j] = mult * base^(j << log_stride)
// (counterpart of PrecomputedOmegas::new_for_domain, src/precomputations/mod.rs:14-66)
// fmt 0: 32-byte R-form entries (fri.hip, pointwise.hip); fmt 1: 48-byte 9 x 29-bit entries in
// R' = 2^261 form (value * 2^5), the multiplier operand of fr9_mul.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_pow_table(uint4 *out, Fr base, Fr mult, uint32_t log_stride, uint64_t count, uint32_t fmt, FrParams P BXPARAM)
{
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    Fr w = fr_pow(base, j << log_stride, P);
    w = fr_mul(w, mult, P);
    if (fmt == 0) {
        fr_store(BATS(1, out, 2 * j, 2), w);
    } else {
#pragma unroll
        for (int i = 0; i < 5; i++) w = fr_add(w, w, P);   // * 2^5: R-form -> R'-form
        fr9_store48(BATS(2, out, 3 * j, 3), fr9_unpack(w));
    }
}

// W3 entries (fr9w3.cuh): out[j] = { v 2^87, v 2^174, v 2^261 } mod p for the plain integer
// v = mult * base^(j << log_stride).  K.k[c] holds 2^(87 (c + 1)) mod p as a plain integer, so the
// Montgomery product with the R-form power strips the R and leaves the plain, canonical W_c.
__global__ void __launch_bounds__(256)
k_pow_table_w3(uint4 *out, Fr base, Fr mult, uint32_t log_stride, uint64_t count, W3Consts K, FrParams P BXPARAM)
{
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    Fr w = fr_pow(base, j << log_stride, P);
    w = fr_mul(w, mult, P);
    Fr9W3 e;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        Fr9 t = fr9_unpack(fr_mul(w, K.k[c], P));
#pragma unroll
        for (int i = 0; i < 9; i++) e.w[c][i] = t.v[i];
    }
    fr9w3_store(BATS(1, out, 7 * j, 7), e);
}

// W9 entries (fr9w3.cuh): out[j] = V[k][c] = limb k of ( v 2^(29 (c + 1)) mod p ), column-major with 12 words per
// column, for the plain integer v = base^(j << log_stride).
__global__ void __launch_bounds__(64)
k_pow_table_w9(uint32_t *out, Fr base, uint32_t log_stride, uint32_t count, W9Consts K, FrParams P BXPARAM)
{
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    Fr w = fr_pow(base, (uint64_t)j << log_stride, P);
    uint32_t *e = BATS(1, out, (size_t)j * W9_WORDS, W9_WORDS);
    for (int c = 0; c < 9; c++) {
        Fr9 t = fr9_unpack(fr_mul(w, K.k[c], P));
        for (int k = 0; k < 9; k++) e[12 * k + c] = t.v[k];
    }
    for (int k = 0; k < 9; k++)
        for (int c = 9; c < 12; c++) e[12 * k + c] = 0;
}

// ---------------------------------------------------------------------------------------------
// LDS element accessors: limbs 0-3 at a4[s], limbs 4-7 at b4[s], limb 8 at c1[s]
// ---------------------------------------------------------------------------------------------
struct LdsView {
    uint4 *a4;
    uint4 *b4;
    uint32_t *c1;
};

__device__ __forceinline__ Fr9 lds_get(const LdsView &L, uint32_t s)
{
    uint4 lo = L.a4[s], hi = L.b4[s];
    Fr9 r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    r.v[8] = L.c1[s];
    return r;
}

__device__ __forceinline__ void lds_put(const LdsView &L, uint32_t s, const Fr9 &a)
{
    L.a4[s] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    L.b4[s] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
    L.c1[s] = a.v[8];
}

// a - b + 11p for subtrahends < 10p (the W9 products)
__device__ __forceinline__ Fr9 fr9_sub11(const Fr9 &a, const Fr9 &b, const Fr9Params &P)
{
    Fr9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + (P.c11p[i] - b.v[i]);
    return r;
}

// a - b + 5p, limb-wise non-negative when b is normalized and < 4p (c5p: 5p with limbs 0..7 raised
// by 2^29 borrowed from the limb above, so its top limb still exceeds that of any value < 4p)
__device__ __forceinline__ Fr9 fr9_sub5(const Fr9 &a, const Fr9 &b, const Fr9Params &P)
{
    Fr9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + (P.c5p[i] - b.v[i]);
    return r;
}

// x * base^e through the two-level W3 table: base^e = hi[e >> lo_bits] * lo[e & mask], applied as two
// successive data x constant products (x normalized on entry, result normalized and < 4p).
// `always`: hi[0] is not 1 (it carries the folded n^-1 scale), so it is applied even for exponent 0.
template <bool P1>
__device__ __forceinline__ Fr9 mul_two_level(Fr9 x, const TwoLevel &t, uint64_t e, bool always, const Fr9Params &Q BXPARAM)
{
    const uint64_t lo_i = e & ((1ull << t.lo_bits) - 1), hi_i = e >> t.lo_bits;
    if (hi_i != 0 || always) x = fr9_mul3<P1>(x, fr9w3_load(BAT(20, t.hi, 7 * hi_i, 7)), Q);
    if (lo_i != 0) x = fr9_mul3<P1>(x, fr9w3_load(BAT(21, t.lo, 7 * lo_i, 7)), Q);
    return x;
}

// ---------------------------------------------------------------------------------------------
// K3: one Stockham pass.  grid = (n / (R*C), batch) workgroups.
//
//   input  index  j + i * (n/R)        i < R (sub-transform input), j < n/R
//   twiddle       w_(L*R)^(i*p)        p = j mod L, L = product of earlier radices
//   output index  (j - p)*R + p + c*L  c < R (sub-transform output)
// ---------------------------------------------------------------------------------------------
#ifdef HODOR_TWOPASS   // experiment build (bench/twopass.sh): 4096-point tiles, one workgroup of 1024 threads per CU
constexpr int NTT_MAX_THREADS = 1024;
#else
constexpr int NTT_MAX_THREADS = 512;
#endif

// HODOR_ABLATE builds (csrc/Makefile target `ablate`, bench/ablate.sh) carry the phase-skipping
// switches used to apportion the kernel's time; the shipped library is compiled without them.
#ifdef HODOR_ABLATE
#define ABL(bit) (A.dbg & (bit))
// HODOR_DBG bit 32: intermediates cross HBM without reduction / pack / unpack (the arithmetic a lazy 48-byte
// record format would save, at unchanged traffic: an upper bound for that design).  HODOR_DBG bit 16: wave 0 of every 64th workgroup stamps the shader clock at its phase boundaries into
// hodor_ablate_stamps (read back by bench/phase_timeline.py); bits 8-9 select the pass (log_l / 8)
__device__ unsigned long long hodor_ablate_stamps[1024 * 16];
#define STAMP(slot)                                                                                   \
    do {                                                                                              \
        if ((A.dbg & 16) && ((A.dbg >> 8) & 3) == (A.log_l >> 3) && threadIdx.x == 0 &&               \
            (blockIdx.x & 63) == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (blockIdx.x >> 6) < 1024)                    \
            hodor_ablate_stamps[(blockIdx.x >> 6) * 16 + (slot)] = __builtin_readcyclecounter();      \
    } while (0)
#else
#define ABL(bit) false
#define STAMP(slot) do {} while (0)
#endif

__device__ __forceinline__ uint64_t split_index(const SplitAddr &S, uint64_t x, uint32_t b)
{
    return (x >> S.hi_log) * S.stride_hi + ((x >> S.lo_log) & S.mid_mask) * S.stride_mid + b * S.batch_stride +
           (x & ((1ull << S.lo_log) - 1));
}

// MODE 0: plain arrays (every transform of the single-GPU API).  MODE 1: the same pass with the
// generalized layouts of PassArgs (column mode, 2D twiddle, split addressing) compiled in.
// P1: modulus = 1 mod 2^29 (fr9_mont_digit): chosen by the launcher from Fr9Params::pinv.
// All arguments of the pass as ONE kernel argument, so that the store phase can read its own from the kernel-argument
// segment when it gets there (`late` below) instead of holding them in SGPRs (spilled to VGPR lanes) through the whole pass.
struct PassKArgs {
    PassArgs A;
    Fr9 scale;
    uint32_t has_scale;
    Fr9Params Q;
#ifdef HODOR_BOUNDS
    BX X;                      // the extents of this pass's buffers (bounds.cuh)
#endif
};
typedef const __attribute__((address_space(4))) PassKArgs *PassKArgsLate;

template <int MODE, bool P1>
__global__ void __launch_bounds__(NTT_MAX_THREADS) __attribute__((amdgpu_waves_per_eu(4)))
k_ntt_pass(PassKArgs K)
{
    const PassArgs &A = K.A;
    const Fr9Params &Q = K.Q;
#ifdef HODOR_BOUNDS
    const BX &X = K.X;
#endif
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
    const uint32_t log_r = A.log_r, log_c = A.log_c;
    const uint32_t R = 1u << log_r, C = 1u << log_c;
    // element (row, c) sits at slot row*C + (c ^ (row & (C-1))): the XOR swizzle spreads a column over
    // all C 16-byte bank groups without the 1/C padding (four workgroups must fit the 160 KiB of a CU)
    // ... and rows 16 apart — what the lanes of a wave-uniform step (below) differ by — are spread over the bank
    // groups by XOR-ing row bits 5:4 into bits 1:0 of the row's place (R >= 64 only)
    const uint32_t Cm = C - 1;
    const uint32_t Rsw = log_r >= 6 ? 3u : 0u;
#define SLOT(row, c) (((((uint32_t)(row)) ^ ((((uint32_t)(row)) >> 4) & Rsw)) << log_c) + (((uint32_t)(c)) ^ (((uint32_t)(row)) & Cm)))
    const uint32_t slots = R << log_c;
    const uint32_t half_r = (R >> 1) ? (R >> 1) : 1;
    // carve: data a4 | data b4 | twiddles (7 x 16 B per W3 entry) | data c1
    LdsView D;
    D.a4 = smem;
    D.b4 = smem + slots;
    uint4 *const T = smem + 2 * slots;
    // tw_sub = s > 0: only every 2^s-th twiddle is staged (all that the steps before the last one touch:
    // their table indices are multiples of 4); the last radix-4 step, the one with R/2 distinct twiddles,
    // reads its W3 entries from the L1/L2-resident global table instead.  A quarter of the LDS table buys
    // a fourth (R = 256) or third (R = 512) resident workgroup per CU.
    const uint32_t tw_sub = A.tw_sub;
    const uint32_t tw_entries = half_r >> tw_sub ? half_r >> tw_sub : 1;
    D.c1 = reinterpret_cast<uint32_t *>(T + 7 * tw_entries);

    STAMP(0);
    // The load phase (table staging, element requests, pre-scale / inter-pass products) runs at raised wave
    // priority: a wave that has not yet put its part of the tile into LDS holds up its whole workgroup at the
    // first barrier, while the other workgroups' butterfly steps on the same SIMD can wait (-3 % on the 2^24
    // step; raising any later phase, or every phase by a different amount, measured neutral or worse).
    __builtin_amdgcn_s_setprio(1);
    // stage omega_R^e (e < R/2, e a multiple of 2^tw_sub) into LDS
    for (uint32_t e = tid; e < 7 * tw_entries; e += nthreads) {
        uint32_t ent = e / 7, q = e - 7 * ent;
        T[LAT(31, e, 7 * tw_entries)] = *BAT(1, A.rtw, 7 * (ent << tw_sub) + q, 1);
    }

    // batched transforms: grid.y (continued in grid.z beyond 65535, see ntt_launch_pass) selects one of `batch`
    // independent size-n arrays — or, in column mode, the tile's first array column
    const uint32_t by = blockIdx.z * gridDim.y + blockIdx.y;
    const uint4 *src_b = A.src + 2ull * by * A.src_batch_stride;   // first LDE pass: n/f apart
    const uint64_t n_over_r = 1ull << (A.log_n - log_r);
    const bool colm = MODE == 1 && A.col_mode;
    const uint64_t colbase = colm ? ((uint64_t)by << log_c) : 0;   // first array column of this tile
    const uint64_t Lmask = (1ull << A.log_l) - 1;
    const uint32_t tw_shift = A.log_n - A.log_l - log_r;   // exponent scale N / (L*R)
    const uint32_t tile = R << log_c;
    const uint64_t j0 = colm ? (uint64_t)blockIdx.x : (uint64_t)blockIdx.x << log_c;

    // ---- load: global -> (pre-scale, inter-pass twiddle) -> LDS at bit-reversed row
    // Zero padding (LDE, src/fft/lde.rs:28-31 `is_non_zero`): when only the first n >> s inputs are
    // non-zero, only sub-transform inputs i < R >> s are, they land on rows that are multiples of 2^s,
    // and the first s radix-2 stages merely copy each of them to the 2^s rows of its group — so load
    // 1/2^s of the tile, replicate, and start the butterflies at stage s.
    const uint32_t log_skip = A.log_skip;
    for (uint32_t e = tid; e < (tile >> log_skip); e += nthreads) {
        uint32_t c = e & (C - 1), i = e >> log_c;
        uint64_t j = colm ? j0 : j0 + c;
        uint64_t g = j + (uint64_t)i * n_over_r;
        Fr9 x;
        if (g < A.nnz) {
            if (ABL(4)) {
#pragma unroll
                for (int k = 0; k < 9; k++) x.v[k] = ((uint32_t)g + k) & HODOR_M29;
            } else if (MODE == 1 && colm) {
                // the row of a column-mode source may itself arrive cut into slabs / chunks (src_split)
                const uint64_t srow = A.src_split.on ? split_index(A.src_split, g, 0) : g;
                x = fr9_unpack(fr_load(BAT(2, A.src, 2 * ((srow << A.src_log_width) + A.src_col_off + colbase + c), 2)));
                if (A.tw2d.lo != nullptr && A.tw2d_on_load) x = mul_two_level<P1>(x, A.tw2d, g * (A.col0 + colbase + c), false, Q BXPASS);
            } else if (MODE == 1 && A.src_split.on) {
                x = fr9_unpack(fr_load(BAT(3, A.src, 2 * split_index(A.src_split, g, by), 2)));
            } else if (ABL(32) && A.apply_tw) {   // upper bound of "lazy intermediates": the words as they are, no unpack
                Fr raw = fr_load(BATP(4, A.src, 2ull * by * A.src_batch_stride + 2 * g, 2, src_b + 2 * g));
#pragma unroll
                for (int k = 0; k < 8; k++) x.v[k] = raw.v[k] & HODOR_M29;
                x.v[8] = 0;
            } else {
                x = fr9_unpack(fr_load(BATP(4, A.src, 2ull * by * A.src_batch_stride + 2 * g, 2, src_b + 2 * g)));
            }
            if (A.pre.lo != nullptr) x = mul_two_level<P1>(x, A.pre, g, false, Q BXPASS);
            if (A.apply_tw && !ABL(2)) {
                uint64_t ex = ((uint64_t)i * (j & Lmask)) << tw_shift;
                x = mul_two_level<P1>(x, A.tw, ex, A.tw_always != 0, Q BXPASS);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 9; k++) x.v[k] = 0;
        }
        uint32_t row = log_r ? (__brev(i) >> (32 - log_r)) : 0u;
        for (uint32_t d = 0; d < (1u << log_skip); d++) lds_put(D, LAT(30, SLOT(row + d, c), slots), x);
    }
    STAMP(1);
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    STAMP(2);

    // ---- R-point DIT in LDS (values lazily reduced: limbs re-normalized once per step)
    uint32_t log_m = log_skip;
    if ((log_r - log_m) & 1) {   // odd number of stages left: one radix-2 stage at half-size m
        const uint32_t m = 1u << log_m;
        const uint32_t items = (R >> 1) << log_c;
        for (uint32_t w = tid; w < items; w += nthreads) {
            uint32_t c = w & (C - 1), q = w >> log_c;
            uint32_t jp = q & (m - 1);
            uint32_t r0 = ((q >> log_m) << (log_m + 1)) + jp;
            uint32_t s0 = SLOT(r0, c), s1 = SLOT(r0 + m, c);
            Fr9 x0 = lds_get(D, LAT(30, s0, slots)), x1 = lds_get(D, LAT(30, s1, slots));
            // m == 1: twiddle 1 and x1 is a stored value (normalized, < 4p), a valid subtrahend
            if (m > 1) x1 = fr9_mul3<P1>(x1, fr9w3_load(T + 7 * LAT(32, (jp << (log_r - log_m - 1)) >> tw_sub, tw_entries)), Q);
            Fr9 y0 = fr9_add(x0, x1), y1 = fr9_sub5(x0, x1, Q);
            fr9_normalize(y0);
            fr9_normalize(y1);
            lds_put(D, LAT(30, s0, slots), y0);
            lds_put(D, LAT(30, s1, slots), y1);
        }
        log_m += 1;
        __syncthreads();
    }
    if (ABL(1)) log_m = log_r;
    // Lazy hand-over between steps (round 4): every radix-4 step but the last one of the pass stores its four sums
    // WITHOUT the carry propagation (limbs < 5 * 2^29: a normalized x0 plus at most two offset subtractions, < 2^30
    // each, and two products, < 2^29 each), and the next step brings in what it needs itself.  A W3 step carries x0 —
    // only ever added to — in full, so that its own sums stay below 5 * 2^29 again; x1 and x3 enter products, which
    // need their 87-bit limb GROUPS below 2^87 (fr9_normalize_groups, 6 instructions); x2 enters two additions (limbs
    // < 7 * 2^29 < 2^32) and is group-carried with the rest of the mid-step sums.  Column sums of those products: at
    // most seven limbs < 7 * 2^29 and two < 2^29 times a table limb < 2^29, plus three reduction terms and the carry:
    // < 55 * 2^58 < 2^64 (replayed with the worst limbs in tests/test_lazy_step_bounds_cpu.py).  Per item 27 + 4 x 6
    // carry instructions instead of 4 x 27 + 2 x 6.  A wave-uniform (W9) step multiplies limb by limb and carries x0,
    // x1 and x3 in full (3 x 27 instead of 4 x 27).  The store phase takes normalized limbs: the last step carries its
    // sums.  Which form a step has is decided at COMPILE time (the W3 step exists with and without the final carry, a
    // W9 step is never the last one): with run-time flags the extra live SGPRs and branches cost what the carries save
    // (profiles/r04/lazy_ab.txt).  -DHODOR_NO_LAZY builds the round-3 hand-over for A/B runs.
#ifdef HODOR_NO_LAZY
    constexpr bool LAZY = false;
#else
    constexpr bool LAZY = true;
#endif
    const uint32_t first_lm = log_m;   // the step that reads what the load phase (or the radix-2 stage) stored: normalized
    for (; log_m < log_r; log_m += 2) {   // radix-4 step = stages with half-size m and 2m
        const uint32_t m = 1u << log_m;
        const uint32_t items = (R >> 2) << log_c;
        if (log_m < A.w9_limit) {
            // ---- wave-uniform step: the step's twiddles take only m (<= 8) different sets of values, and the
            // items are dealt so that the 64 lanes of a wave share ONE of them (jp = twiddle index): the constants
            // then come from scalar loads and enter the multiplier as SGPR operands in the W9 form — 91 multiplier
            // slots per product instead of 111 (fr9w3.cuh).  Chunk g of 64 items: jp = g / (chunks per jp), rotated
            // by the workgroup index so that the cheap jp = 0 (w9_skip_one) lands on every wave slot in turn.
            const uint32_t lane = tid & 63, wave = tid >> 6, waves = nthreads >> 6;
            const uint32_t log_chunks = log_r - 2 + log_c - 6;          // items / 64
            const uint32_t log_cpj = log_chunks - log_m;               // chunks per twiddle index
            const W9Ptr W9 = (W9Ptr)A.rtw9;
            const uint32_t e_shift = log_r - 5;                        // table index = exponent / (R / 32)
            for (uint32_t g = wave; g < (1u << log_chunks); g += waves) {
                const uint32_t jp = __builtin_amdgcn_readfirstlane(((g >> log_cpj) + blockIdx.x + by) & (m - 1));
                const uint32_t sub = g & ((1u << log_cpj) - 1);
                const uint32_t c = lane & (C - 1), kb = (sub << (6 - log_c)) + (lane >> log_c);
                const uint32_t r0 = (kb << (log_m + 2)) + jp;
                const uint32_t s0 = SLOT(r0, c), s1 = SLOT(r0 + m, c), s2 = SLOT(r0 + 2 * m, c),
                               s3 = SLOT(r0 + 3 * m, c);
                Fr9 x0 = lds_get(D, LAT(30, s0, slots));
                Fr9 x1 = lds_get(D, LAT(30, s1, slots));
                Fr9 x2 = lds_get(D, LAT(30, s2, slots));
                Fr9 x3 = lds_get(D, LAT(30, s3, slots));
                Fr9 t;
                if (LAZY && log_m != first_lm) {
                    fr9_normalize(x0);
                    fr9_normalize(x1);
                    fr9_normalize(x3);
                }
                const bool ones = jp == 0 && (m == 1 || A.w9_skip_one);   // wa = wb = 1 (wave-uniform branch)
                if (!ones) {
                    (void)BAT(6, A.rtw9, W9_WORDS * ((jp << (log_r - log_m - 1)) >> e_shift), W9_WORDS);
                    fr9_mul9x2<P1>(x1, x3, W9 + W9_WORDS * __builtin_amdgcn_readfirstlane((jp << (log_r - log_m - 1)) >> e_shift), Q);
                    t = x1; x1 = fr9_sub11(x0, t, Q); x0 = fr9_add(x0, t);
                    t = x3; x3 = fr9_sub11(x2, t, Q); x2 = fr9_add(x2, t);
                    fr9_normalize(x2);
                    fr9_normalize(x3);
                    (void)BAT(7, A.rtw9, W9_WORDS * ((jp << (log_r - log_m - 2)) >> e_shift), W9_WORDS);
                    t = fr9_mul9<P1>(x2, W9 + W9_WORDS * __builtin_amdgcn_readfirstlane((jp << (log_r - log_m - 2)) >> e_shift), Q);
                    x2 = fr9_sub11(x0, t, Q); x0 = fr9_add(x0, t);
                } else {
                    // twiddles 1: the subtrahends are stored values, brought under 2p first unless this is the
                    // pass's first step (then they are loaded values, normalized and < 4p)
                    if (m > 1) { fr9_reduce_partial<true>(x1, Q); fr9_reduce_partial<true>(x3, Q); }
                    t = x1; x1 = fr9_sub5(x0, t, Q); x0 = fr9_add(x0, t);
                    t = x3; x3 = fr9_sub5(x2, t, Q); x2 = fr9_add(x2, t);
                    t = x2;
                    fr9_reduce_partial(t, Q);
                    x2 = fr9_sub5(x0, t, Q); x0 = fr9_add(x0, t);
                    fr9_normalize(x3);
                }
                (void)BAT(8, A.rtw9, W9_WORDS * (((jp + m) << (log_r - log_m - 2)) >> e_shift), W9_WORDS);
                t = fr9_mul9<P1>(x3, W9 + W9_WORDS * __builtin_amdgcn_readfirstlane(((jp + m) << (log_r - log_m - 2)) >> e_shift), Q);
                x3 = fr9_sub11(x1, t, Q); x1 = fr9_add(x1, t);
                if (!LAZY) {   // (a W9 step is never the last step of a pass: ntt_launch_pass)
                    fr9_normalize(x0);
                    fr9_normalize(x1);
                    fr9_normalize(x2);
                    fr9_normalize(x3);
                }
                lds_put(D, LAT(30, s0, slots), x0);
                lds_put(D, LAT(30, s1, slots), x1);
                lds_put(D, LAT(30, s2, slots), x2);
                lds_put(D, LAT(30, s3, slots), x3);
            }
            STAMP(3 + log_m);
            __syncthreads();
            STAMP(4 + log_m);
            continue;
        }
        const bool tw_global = tw_sub != 0 && log_r - log_m - 2 < tw_sub;   // finest index of this step: jp << (log_r - log_m - 2)
        auto step_items = [&](auto from_global, auto carry_out) {
            constexpr bool G = decltype(from_global)::value;
            constexpr bool CARRY_OUT = decltype(carry_out)::value;   // last step of the pass (or a HODOR_NO_LAZY build)
            auto twiddle = [&](uint32_t idx) -> Fr9W3 {
                if constexpr (G) return fr9w3_load(BAT(5, A.rtw, 7 * idx, 7));
                else return fr9w3_load(T + 7 * LAT(32, idx >> tw_sub, tw_entries));
            };
            for (uint32_t w = tid; w < items; w += nthreads) {
                uint32_t c = w & (C - 1), q = w >> log_c;
                uint32_t jp = q & (m - 1);
                uint32_t k = (q >> log_m) << (log_m + 2);
                const uint32_t r0 = k + jp;
                const uint32_t s0 = SLOT(r0, c), s1 = SLOT(r0 + m, c), s2 = SLOT(r0 + 2 * m, c),
                               s3 = SLOT(r0 + 3 * m, c);
                Fr9 x0 = lds_get(D, LAT(30, s0, slots));
                Fr9 x1 = lds_get(D, LAT(30, s1, slots));
                Fr9 x2 = lds_get(D, LAT(30, s2, slots));
                Fr9 x3 = lds_get(D, LAT(30, s3, slots));
                Fr9 t;
                if (m > 1) {
                    if (LAZY) {   // (also when this is the pass's first radix-4 step and the limbs are normalized already)
                        fr9_normalize(x0);
                        fr9_normalize_groups(x1);
                        fr9_normalize_groups(x3);
                    }
                    const Fr9W3 wa = twiddle(jp << (log_r - log_m - 1));
#ifdef HODOR_EXP_X2
                    {
                        Fr9 t1 = x1, t3 = x3;
                        fr9_mul3x2<P1>(t1, t3, wa, Q);
                        x1 = fr9_sub5(x0, t1, Q); x0 = fr9_add(x0, t1);
                        x3 = fr9_sub5(x2, t3, Q); x2 = fr9_add(x2, t3);
                    }
#else
                    t = fr9_mul3<P1>(x1, wa, Q);
                    x1 = fr9_sub5(x0, t, Q); x0 = fr9_add(x0, t);
                    t = fr9_mul3<P1>(x3, wa, Q);
                    x3 = fr9_sub5(x2, t, Q); x2 = fr9_add(x2, t);
#endif
                    // the second stage multiplies the lazy sums (limbs < 2^29 + 2^30): the W3 product only needs
                    // its three 87-bit limb GROUPS below 2^87 for its (4 + 2^-25) p bound — two carries, not eight
#ifdef HODOR_EXP_FULLNORM
                    fr9_normalize(x2);
                    fr9_normalize(x3);
#else
                    fr9_normalize_groups(x2);
                    fr9_normalize_groups(x3);
#endif
                    t = fr9_mul3<P1>(x2, twiddle(jp << (log_r - log_m - 2)), Q);
                    x2 = fr9_sub5(x0, t, Q); x0 = fr9_add(x0, t);
                } else {
                    // twiddles are 1: subtrahends are stored values (normalized, < 4p), except the
                    // stage-B one, a lazy sum that is first brought back under 2p
                    t = x1; x1 = fr9_sub5(x0, t, Q); x0 = fr9_add(x0, t);
                    t = x3; x3 = fr9_sub5(x2, t, Q); x2 = fr9_add(x2, t);
                    t = x2;
                    fr9_reduce_partial(t, Q);
                    x2 = fr9_sub5(x0, t, Q); x0 = fr9_add(x0, t);
                    fr9_normalize(x3);
                }
                t = fr9_mul3<P1>(x3, twiddle((jp + m) << (log_r - log_m - 2)), Q);
                x3 = fr9_sub5(x1, t, Q); x1 = fr9_add(x1, t);
                if (CARRY_OUT) {
                    fr9_normalize(x0);
                    fr9_normalize(x1);
                    fr9_normalize(x2);
                    fr9_normalize(x3);
                }
                lds_put(D, LAT(30, s0, slots), x0);
                lds_put(D, LAT(30, s1, slots), x1);
                lds_put(D, LAT(30, s2, slots), x2);
                lds_put(D, LAT(30, s3, slots), x3);
            }
        };
        // three forms: a step that is followed by another one reads the LDS twiddle table (its indices are multiples
        // of 2^tw_sub, see the launcher) and stores lazy sums; the last step reads the LDS or the global table
        if (LAZY && log_m + 2 < log_r && !tw_global) step_items(std::false_type{}, std::false_type{});
        else if (tw_global) step_items(std::true_type{}, std::true_type{});
        else step_items(std::false_type{}, std::true_type{});
        STAMP(3 + log_m);          // 3, 5, 7, 9: end of the arithmetic of the step with half-size 2^log_m
        __syncthreads();
        STAMP(4 + log_m);          // 4, 6, 8, 10: released from its barrier
    }

    // ---- store: LDS -> (scale, post-scale) -> reduce -> global, Stockham output index
    // What only this phase needs (destination, output scalings, the reduction's constants, the 4-step layouts) is read
    // from the kernel-argument segment HERE: the compiler loads by-value kernel arguments in the entry block, and with
    // the SGPR file full through the butterfly steps those that are used last spend the pass in VGPR lanes
    // (v_writelane / v_readlane).  The empty asm makes the segment's address opaque, so these loads cannot be merged
    // with (or hoisted to) the entry block's.
#ifdef HODOR_NO_LATE_ARGS
    const PassKArgs *const late = &K;
#else
    uint64_t kseg = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kseg));
    const PassKArgsLate late = (PassKArgsLate)kseg;
#endif
    const uint32_t s_log_l = late->A.log_l, s_log_n = late->A.log_n;
    const uint64_t s_lmask = (1ull << s_log_l) - 1;
    uint4 *const s_dst = late->A.dst;
    uint4 *const s_dst_b = s_dst + ((2ull * by) << s_log_n);
    const TwoLevel s_post = {late->A.post.lo, late->A.post.hi, late->A.post.lo_bits};
    Fr9Params QS;
#pragma unroll
    for (int k = 0; k < 9; k++) { QS.p[k] = Q.p[k]; QS.c4p[k] = 0; QS.c5p[k] = 0; QS.c11p[k] = 0; }
    QS.pinv = Q.pinv;
    QS.mu = late->Q.mu;
    QS.red_shift = late->Q.red_shift;
    const bool s_has_scale = late->has_scale != 0;
    Fr9 s_scale;
#pragma unroll
    for (int k = 0; k < 9; k++) s_scale.v[k] = s_has_scale ? late->scale.v[k] : 0u;
    const bool transposed = (s_log_l == 0) && !colm;   // first pass: outputs of one sub-transform are contiguous
    const bool last = (s_log_l + log_r == s_log_n);
    for (uint32_t e = tid; e < tile; e += nthreads) {
        uint32_t c, cc;
        if (transposed) { cc = e & (R - 1); c = e >> log_r; }
        else            { c = e & (C - 1);  cc = e >> log_c; }
        uint64_t j = colm ? j0 : j0 + c;
        uint64_t p = j & s_lmask;
        uint64_t o = ((j - p) << log_r) + p + ((uint64_t)cc << s_log_l);
        Fr9 x = lds_get(D, LAT(30, SLOT(cc, c), slots));
        if (s_has_scale) x = fr9_mul(x, s_scale, QS);
        if (s_post.lo != nullptr) x = mul_two_level<P1>(x, s_post, o, false, QS BXPASS);
        if (MODE == 1 && colm && late->A.tw2d.lo != nullptr && !late->A.tw2d_on_load) {
            const TwoLevel t2 = {late->A.tw2d.lo, late->A.tw2d.hi, late->A.tw2d.lo_bits};
            x = mul_two_level<P1>(x, t2, o * (late->A.col0 + colbase + c), false, QS BXPASS);
        }
        // x is normalized here: either straight from LDS (carry-propagated by the last step) or a product
        Fr y;
        if (ABL(32) && !last) {                   // ... and no reduction / pack on the way out
#pragma unroll
            for (int k = 0; k < 8; k++) y.v[k] = x.v[k];
        } else {
            y = last ? fr9_to_canonical<true>(x, QS) : fr9_to_packed<true>(x, QS);
        }
        if (ABL(8) && y.v[0] != 0x12345u) continue;
        // streaming stores: the output crosses the chip once and should not push the twiddle tables out of L2
        // (-1 % on the 2^24 step; streaming LOADS of the data measured +0.8 %)
        if (MODE == 1 && late->A.peer_tab != nullptr) {
            // direct exchange: the slab this element belongs to lives in another rank's receive buffer
            uint64_t t, at;
            const uint32_t peer_self = late->A.peer_self;
            if (colm) {
                const uint32_t peer_log = late->A.peer_log;
                t = o >> peer_log;
                const uint64_t row = ((uint64_t)peer_self << peer_log) + (o & ((1ull << peer_log) - 1));
                at = (row << late->A.dst_log_width) + late->A.dst_col_off + colbase + c;
            } else {
                const uint32_t hi_log = late->A.dst_split.hi_log, lo_log = late->A.dst_split.lo_log;
                t = o >> hi_log;
                at = peer_self * late->A.dst_split.stride_hi + ((o >> lo_log) & late->A.dst_split.mid_mask) * late->A.dst_split.stride_mid +
                     by * late->A.dst_split.batch_stride + (o & ((1ull << lo_log) - 1));
            }
            uint4 *base = reinterpret_cast<uint4 *>(*BAT(12, late->A.peer_tab, t, 1));
            fr_store_nt(BATS(13, base, 2 * (late->A.peer_off + at), 2), y);
        } else if (MODE == 1 && colm) fr_store_nt(BATS(14, s_dst, 2 * ((o << late->A.dst_log_width) + late->A.dst_col_off + colbase + c), 2), y);
        else if (MODE == 1 && late->A.dst_split.on) {
            SplitAddr S;
            S.on = 1; S.lo_log = late->A.dst_split.lo_log; S.hi_log = late->A.dst_split.hi_log; S.mid_mask = late->A.dst_split.mid_mask;
            S.stride_mid = late->A.dst_split.stride_mid; S.stride_hi = late->A.dst_split.stride_hi; S.batch_stride = late->A.dst_split.batch_stride;
            fr_store_nt(BATS(15, s_dst, 2 * split_index(S, o, by), 2), y);
        } else fr_store_nt(BATSP(16, s_dst, ((2ull * by) << s_log_n) + 2 * o, 2, s_dst_b + 2 * o), y);
    }
    STAMP(11);
}

// ---------------------------------------------------------------------------------------------
// host-side launch helpers (called from the C-ABI layer)
// ---------------------------------------------------------------------------------------------
size_t ntt_pass_lds_bytes(uint32_t log_r, uint32_t log_c, uint32_t tw_sub)
{
    size_t R = (size_t)1 << log_r, C = (size_t)1 << log_c;
    size_t half_r = (R / 2) ? R / 2 : 1;
    size_t entries = (half_r >> tw_sub) ? half_r >> tw_sub : 1;
    return R * C * 36 + entries * 112;
}

hipError_t ntt_launch_pass(hipStream_t stream, const PassArgs &A, const Fr9 *scale, const Fr9Params &Q)
{
    static const hipError_t attr_rc = [] {
        const void *fns[4] = {reinterpret_cast<const void *>(k_ntt_pass<0, false>), reinterpret_cast<const void *>(k_ntt_pass<0, true>),
                              reinterpret_cast<const void *>(k_ntt_pass<1, false>), reinterpret_cast<const void *>(k_ntt_pass<1, true>)};
        for (const void *f : fns) {
            hipError_t rc = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (rc != hipSuccess) return rc;
        }
        return hipSuccess;
    }();
    if (attr_rc != hipSuccess) return attr_rc;
    uint64_t n = 1ull << A.log_n;
    const bool general = A.col_mode || A.src_split.on || A.dst_split.on || A.peer_tab != nullptr;
    // column mode: one sub-transform position per workgroup, grid.y walks the array's columns C at a time
    uint64_t grid = A.col_mode ? n >> A.log_r : n >> (A.log_r + A.log_c);
    // grid.y is limited to 65535: a larger count (2^16 tile-column groups of a wide column-mode array, 2^16+ rows of
    // a 4-step row block) continues in grid.z — the kernel reads blockIdx.z * gridDim.y + blockIdx.y
    uint64_t grid_y = A.col_mode ? (1ull << (A.log_width - A.log_c)) : (A.batch ? A.batch : 1);
    unsigned grid_z = 1;
    while (grid_y > 65535) {
        if ((grid_y & 1) || grid_z >= 32768) return hipErrorInvalidValue;
        grid_y >>= 1;
        grid_z <<= 1;
    }
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    Fr9 s = {};
    if (scale) s = *scale;
    // sub-sampled LDS twiddle table when it raises the number of resident workgroups: every step but the
    // last indexes multiples of 4 as long as at least two radix-4 steps follow the (optional) radix-2 stage
    PassArgs B = A;
    B.tw_sub = 0;
    {
        const uint32_t stages = A.log_r - A.log_skip;
        const bool eligible = A.log_r >= 6 && stages >= 4 && ((stages & 1) == 0 || A.log_r - A.log_skip - 1 >= 2);
        if (eligible) {
            size_t full = ntt_pass_lds_bytes(A.log_r, A.log_c, 0), sub = ntt_pass_lds_bytes(A.log_r, A.log_c, 2);
            // the knob only switches the optimisation off: a tile whose full table does not fit is thinned anyway
            if ((knobs().ntt_tw_sub || full > 160 * 1024) && (160 * 1024) / sub > (160 * 1024) / full) B.tw_sub = 2;
        }
    }
#ifdef HODOR_TWOPASS
    // a tile that does not fit with a quarter table: keep thinning the LDS table (the steps whose indices fall
    // between its entries read the global one), even-radix un-padded passes only
    while (ntt_pass_lds_bytes(A.log_r, A.log_c, B.tw_sub) > 160 * 1024 && B.tw_sub + 1 < A.log_r && A.log_skip == 0 &&
           (A.log_r & 1) == 0)
        B.tw_sub = B.tw_sub ? B.tw_sub + 1 : 2;
#endif
    // wave-uniform W9 steps (see k_ntt_pass): the radix-4 steps with half-size m <= 2^T, T the largest value for which
    //   - every such step has >= 64 items per twiddle index:  R/4 * C / m >= 64,  and m <= 8 (table granularity R/32),
    //   - the value bound at the end of the pass stays <= 63p (a W9 product is < 10p and is subtracted with an 11p
    //     offset: such a step raises the bound by 22p instead of 10p; 20p instead of 14p for the twiddle-free one).
    B.w9_limit = 0;
    B.w9_skip_one = knobs().ntt_w9 >= 2 ? 1 : 0;
    if (knobs().ntt_w9 && A.rtw9 && A.log_r >= 6) {
        uint32_t lm0 = A.log_skip, bound0 = 4;
        if ((A.log_r - lm0) & 1) { lm0 += 1; bound0 = 9; }
        for (int T = 3; T >= 0; T--) {
            uint32_t bound = bound0;
            bool any = false, ok = true;
            for (uint32_t lm = lm0; lm < A.log_r; lm += 2) {
                const bool w9 = lm <= (uint32_t)T;
                if (w9) {
                    any = true;
                    if (A.log_r - 2 + A.log_c < 6 + lm) ok = false;          // fewer than 64 items per twiddle index
                    if (lm + 2 >= A.log_r) ok = false;                        // never the last step (it stores lazy sums)
                }
                if (lm == 0) bound = w9 ? 20 : 14;
                else bound += w9 ? 22 : 10;
            }
            if (any && ok && bound <= 63) { B.w9_limit = (uint32_t)T + 1; break; }
        }
    }
    size_t lds = ntt_pass_lds_bytes(A.log_r, A.log_c, B.tw_sub);
    if (lds > 160 * 1024) return hipErrorInvalidValue;   // the planner (abi.hip) never asks for such a tile
    // one radix-4 work item per thread when the tile allows it: 512 threads on a 2048-element tile
    const int threads_override = knobs().ntt_threads;
#ifdef HODOR_ABLATE
    static int dbg = -1;
    if (dbg < 0) {
        const char *e = getenv("HODOR_DBG");
        dbg = e ? atoi(e) : 0;
    }
    B.dbg = (uint32_t)dbg;
#endif
    uint32_t items = 1u << (A.log_r + A.log_c >= 2 ? A.log_r + A.log_c - 2 : 0);
    unsigned threads = items >= 512 ? 512 : (items >= 256 ? 256 : (items >= 128 ? 128 : 64));
#ifdef HODOR_TWOPASS
    if (items >= 1024) threads = 1024;
#endif
    // rounded down to whole waves: the wave-uniform steps deal their work per wave (tid >> 6, nthreads >> 6)
    if (threads_override >= 64 && threads_override <= NTT_MAX_THREADS) threads = (unsigned)threads_override & ~63u;
    // p = 1 mod 2^29 <=> -p^-1 = -1 mod 2^29 (Fr9Params::pinv all ones): the Montgomery digit is a negation
    const bool p1 = knobs().ntt_p1 && Q.pinv == HODOR_M29;
    const dim3 g3((unsigned)grid, (unsigned)grid_y, grid_z);
    const uint32_t hs = scale ? 1u : 0u;
    PassKArgs K;
    K.A = B;
    K.scale = s;
    K.has_scale = hs;
    K.Q = Q;
#ifdef HODOR_BOUNDS
    {
        BXB bx(KID_NTT_PASS);
        bx.add(A.src, A.bx_src_bytes).add(A.dst, A.bx_dst_bytes).add(A.rtw, A.bx_rtw_bytes).add(A.rtw9, A.bx_rtw9_bytes);
        bx.add(A.tw.lo, A.tw.lo_bytes).add(A.tw.hi, A.tw.hi_bytes).add(A.pre.lo, A.pre.lo_bytes).add(A.pre.hi, A.pre.hi_bytes);
        bx.add(A.post.lo, A.post.lo_bytes).add(A.post.hi, A.post.hi_bytes).add(A.tw2d.lo, A.tw2d.lo_bytes).add(A.tw2d.hi, A.tw2d.hi_bytes);
        if (A.peer_tab) {
            bx.add(A.peer_tab, A.bx_peers * sizeof(uint64_t));
            for (uint32_t t = 0; t < A.bx_peers; t++) bx.add((const void *)(uintptr_t)A.bx_peer_host[t], A.bx_peer_bytes);
        }
        K.X = bx;
    }
#endif
    if (general && p1) hipLaunchKernelGGL((k_ntt_pass<1, true>), g3, dim3(threads), lds, stream, K);
    else if (general)  hipLaunchKernelGGL((k_ntt_pass<1, false>), g3, dim3(threads), lds, stream, K);
    else if (p1)       hipLaunchKernelGGL((k_ntt_pass<0, true>), g3, dim3(threads), lds, stream, K);
    else               hipLaunchKernelGGL((k_ntt_pass<0, false>), g3, dim3(threads), lds, stream, K);
    return hipGetLastError();
}

hipError_t pow_table_launch(hipStream_t stream, uint4 *out, const Fr &base, const Fr &mult,
                            uint32_t log_stride, uint64_t count, uint32_t fmt, const FrParams &P)
{
    unsigned grid = (unsigned)((count + 255) / 256);
    BX_BEGIN(bx, KID_POW_TABLE);
    BX_ADD(bx, out, count * (fmt ? 48 : 32));
    hipLaunchKernelGGL(k_pow_table, dim3(grid), dim3(256), 0, stream, out, base, mult, log_stride, count, fmt,
                       P BXARG(bx));
    return hipGetLastError();
}

hipError_t pow_table_w9_launch(hipStream_t stream, uint32_t *out, const Fr &base, uint32_t log_stride, uint32_t count,
                               const W9Consts &K, const FrParams &P)
{
    BX_BEGIN(bx, KID_POW_TABLE_W9);
    BX_ADD(bx, out, (size_t)count * W9_WORDS * sizeof(uint32_t));
    hipLaunchKernelGGL(k_pow_table_w9, dim3((count + 63) / 64), dim3(64), 0, stream, out, base, log_stride, count, K, P BXARG(bx));
    return hipGetLastError();
}

hipError_t pow_table_w3_launch(hipStream_t stream, uint4 *out, const Fr &base, const Fr &mult,
                               uint32_t log_stride, uint64_t count, const W3Consts &K, const FrParams &P)
{
    unsigned grid = (unsigned)((count + 255) / 256);
    BX_BEGIN(bx, KID_POW_TABLE_W3);
    BX_ADD(bx, out, count * 112);
    hipLaunchKernelGGL(k_pow_table_w3, dim3(grid), dim3(256), 0, stream, out, base, mult, log_stride, count, K,
                       P BXARG(bx));
    return hipGetLastError();
}

}  // namespace hodor

#ifdef HODOR_ABLATE
extern "C" int hodor_ablate_read_stamps(unsigned long long *host, size_t count)
{
    if (count > 1024 * 16) count = 1024 * 16;
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(hodor::hodor_ablate_stamps), count * sizeof(unsigned long long), 0,
                                    hipMemcpyDeviceToHost);
}
#endif
