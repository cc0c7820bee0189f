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
//   * Elements sit in LDS as two 16-byte halves (lo/hi arrays, rows padded by one slot) so that
//     ds_read_b128 / ds_write_b128 are conflict-free for both row-major and column-major access.
//   * Inter-pass twiddles come from a two-level power table (L2-resident) instead of an n-entry
//     table streamed from HBM.
#include <cstdlib>

#include "ntt.cuh"

namespace hodor {

// ---------------------------------------------------------------------------------------------
// K2: out[j] = mult * base^(j << log_stride)
// (counterpart of PrecomputedOmegas::new_for_domain, src/precomputations/mod.rs:14-66)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_pow_table(uint4 *out, Fr base, Fr mult, uint32_t log_stride, uint64_t count, FrParams P)
{
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    Fr w = fr_pow(base, j << log_stride, P);
    w = fr_mul(w, mult, P);
    fr_store(out + 2 * j, w);
}

// ---------------------------------------------------------------------------------------------
// LDS element accessors: lo halves at [0, slots), hi halves at [slots, 2*slots)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ Fr lds_get(const uint4 *base, uint32_t slots, uint32_t s)
{
    uint4 lo = base[s], hi = base[slots + s];
    Fr r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}

__device__ __forceinline__ void lds_put(uint4 *base, uint32_t slots, uint32_t s, const Fr &a)
{
    base[s] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    base[slots + s] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}

__device__ __forceinline__ Fr two_level_pow(const TwoLevel &t, uint64_t e, const FrParams &P)
{
    uint64_t lo_i = e & ((1ull << t.lo_bits) - 1), hi_i = e >> t.lo_bits;
    Fr h = fr_load(t.hi + 2 * hi_i);
    if (lo_i == 0) return h;
    Fr l = fr_load(t.lo + 2 * lo_i);
    return fr_mul(h, l, P);
}

// ---------------------------------------------------------------------------------------------
// K3: one Stockham pass.  grid = n / (R*C) workgroups of 256 threads.
//
//   input  index  j + i * (n/R)        i < R (sub-transform input), j < n/R
//   twiddle       w_(L*R)^(i*p)        p = j mod L, L = product of earlier radices
//   output index  (j - p)*R + p + c*L  c < R (sub-transform output)
// ---------------------------------------------------------------------------------------------
constexpr int NTT_MAX_THREADS = 512;

__global__ void __launch_bounds__(NTT_MAX_THREADS)
k_ntt_pass(PassArgs A, Fr scale, uint32_t has_scale, FrParams P)
{
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
    const uint32_t log_r = A.log_r, log_c = A.log_c;
    const uint32_t R = 1u << log_r, C = 1u << log_c;
    const uint32_t RS = (C == 1) ? 1u : C + 1;      // padded row stride (slots)
    const uint32_t slots = R * RS;
    uint4 *data = smem;                              // 2 * slots
    uint4 *tw = smem + 2 * slots;                    // 2 * (R/2): sub-transform twiddles
    const uint32_t half_r = R >> 1;

    // stage omega_R^e (e < R/2) into LDS
    for (uint32_t e = tid; e < half_r; e += nthreads) {
        tw[e] = A.rtw[2 * e];
        tw[half_r + e] = A.rtw[2 * e + 1];
    }

    // batched transforms: blockIdx.y selects one of `batch` independent size-n arrays
    const uint4 *src_b = A.src + ((2ull * blockIdx.y) << A.log_n);
    uint4 *dst_b = A.dst + ((2ull * blockIdx.y) << A.log_n);
    const uint64_t n_over_r = 1ull << (A.log_n - log_r);
    const uint64_t j0 = (uint64_t)blockIdx.x << log_c;
    const uint64_t Lmask = (1ull << A.log_l) - 1;
    const uint32_t tw_shift = A.log_n - A.log_l - log_r;   // exponent scale N / (L*R)
    const uint32_t tile = R << log_c;

    // ---- load: global -> (pre-scale, inter-pass twiddle) -> LDS at bit-reversed row
    for (uint32_t e = tid; e < tile; e += nthreads) {
        uint32_t c = e & (C - 1), i = e >> log_c;
        uint64_t j = j0 + c;
        uint64_t g = j + (uint64_t)i * n_over_r;
        Fr x;
        if (g < A.nnz) {
            x = fr_load(src_b + 2 * g);
            if (A.pre.lo != nullptr && g != 0) x = fr_mul(x, two_level_pow(A.pre, g, P), P);
            if (A.apply_tw) {
                uint64_t ex = ((uint64_t)i * (j & Lmask)) << tw_shift;
                if (ex != 0) x = fr_mul(x, two_level_pow(A.tw, ex, P), P);
            }
        } else {
            x = fr_zero();
        }
        uint32_t row = log_r ? (__brev(i) >> (32 - log_r)) : 0u;
        lds_put(data, slots, row * RS + c, x);
    }
    __syncthreads();

    // ---- R-point DIT in LDS
    uint32_t log_m = 0;
    if (log_r & 1) {   // one plain radix-2 stage (all twiddles are 1)
        const uint32_t items = (R >> 1) << log_c;
        for (uint32_t w = tid; w < items; w += nthreads) {
            uint32_t c = w & (C - 1), q = w >> log_c;
            uint32_t s0 = (2 * q) * RS + c, s1 = s0 + RS;
            Fr x0 = lds_get(data, slots, s0), x1 = lds_get(data, slots, s1);
            lds_put(data, slots, s0, fr_add(x0, x1, P));
            lds_put(data, slots, s1, fr_sub(x0, x1, P));
        }
        log_m = 1;
        __syncthreads();
    }
    for (; log_m < log_r; log_m += 2) {   // radix-4 step = stages with half-size m and 2m
        const uint32_t m = 1u << log_m;
        const uint32_t items = (R >> 2) << log_c;
        for (uint32_t w = tid; w < items; w += nthreads) {
            uint32_t c = w & (C - 1), q = w >> log_c;
            uint32_t jp = q & (m - 1);
            uint32_t k = (q >> log_m) << (log_m + 2);
            uint32_t s0 = (k + jp) * RS + c;
            uint32_t st = m * RS;
            Fr x0 = lds_get(data, slots, s0);
            Fr x1 = lds_get(data, slots, s0 + st);
            Fr x2 = lds_get(data, slots, s0 + 2 * st);
            Fr x3 = lds_get(data, slots, s0 + 3 * st);
            Fr t;
            if (m > 1) {
                Fr wa = lds_get(tw, half_r, jp << (log_r - log_m - 1));
                t = fr_mul(x1, wa, P);
                x1 = fr_sub(x0, t, P); x0 = fr_add(x0, t, P);
                t = fr_mul(x3, wa, P);
                x3 = fr_sub(x2, t, P); x2 = fr_add(x2, t, P);
                Fr wb0 = lds_get(tw, half_r, jp << (log_r - log_m - 2));
                t = fr_mul(x2, wb0, P);
                x2 = fr_sub(x0, t, P); x0 = fr_add(x0, t, P);
            } else {
                t = x1; x1 = fr_sub(x0, t, P); x0 = fr_add(x0, t, P);
                t = x3; x3 = fr_sub(x2, t, P); x2 = fr_add(x2, t, P);
                t = x2; x2 = fr_sub(x0, t, P); x0 = fr_add(x0, t, P);
            }
            Fr wb1 = lds_get(tw, half_r, (jp + m) << (log_r - log_m - 2));
            t = fr_mul(x3, wb1, P);
            x3 = fr_sub(x1, t, P); x1 = fr_add(x1, t, P);
            lds_put(data, slots, s0, x0);
            lds_put(data, slots, s0 + st, x1);
            lds_put(data, slots, s0 + 2 * st, x2);
            lds_put(data, slots, s0 + 3 * st, x3);
        }
        __syncthreads();
    }

    // ---- store: LDS -> (scale, post-scale) -> global, Stockham output index
    const bool transposed = (A.log_l == 0);   // first pass: outputs of one sub-transform are contiguous
    for (uint32_t e = tid; e < tile; e += nthreads) {
        uint32_t c, cc;
        if (transposed) { cc = e & (R - 1); c = e >> log_r; }
        else            { c = e & (C - 1);  cc = e >> log_c; }
        uint64_t j = j0 + c;
        uint64_t p = j & Lmask;
        uint64_t o = ((j - p) << log_r) + p + ((uint64_t)cc << A.log_l);
        Fr x = lds_get(data, slots, cc * RS + c);
        if (has_scale) x = fr_mul(x, scale, P);
        if (A.post.lo != nullptr && o != 0) x = fr_mul(x, two_level_pow(A.post, o, P), P);
        fr_store(dst_b + 2 * o, x);
    }
}

// ---------------------------------------------------------------------------------------------
// host-side launch helpers (called from the C-ABI layer)
// ---------------------------------------------------------------------------------------------
size_t ntt_pass_lds_bytes(uint32_t log_r, uint32_t log_c)
{
    size_t R = (size_t)1 << log_r, C = (size_t)1 << log_c;
    size_t RS = (C == 1) ? 1 : C + 1;
    return (2 * R * RS + 2 * (R / 2 ? R / 2 : 1)) * 16;
}

hipError_t ntt_launch_pass(hipStream_t stream, const PassArgs &A, const Fr *scale, const FrParams &P)
{
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_ntt_pass),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    uint64_t n = 1ull << A.log_n;
    uint64_t grid = n >> (A.log_r + A.log_c);
    Fr s = {};
    if (scale) s = *scale;
    size_t lds = ntt_pass_lds_bytes(A.log_r, A.log_c);
    // one radix-4 work item per thread when the tile allows it: 512 threads on a 2048-element tile
    static int threads_override = -1;
    if (threads_override < 0) {
        const char *e = getenv("HODOR_NTT_THREADS");
        threads_override = e ? atoi(e) : 0;
    }
    uint32_t items = 1u << (A.log_r + A.log_c >= 2 ? A.log_r + A.log_c - 2 : 0);
    unsigned threads = items >= 512 ? 512 : (items >= 256 ? 256 : (items >= 128 ? 128 : 64));
    if (threads_override >= 64 && threads_override <= NTT_MAX_THREADS) threads = (unsigned)threads_override;
    hipLaunchKernelGGL(k_ntt_pass, dim3((unsigned)grid, A.batch ? A.batch : 1), dim3(threads), lds, stream, A, s,
                       scale ? 1u : 0u, P);
    return hipGetLastError();
}

hipError_t pow_table_launch(hipStream_t stream, uint4 *out, const Fr &base, const Fr &mult,
                            uint32_t log_stride, uint64_t count, const FrParams &P)
{
    unsigned grid = (unsigned)((count + 255) / 256);
    hipLaunchKernelGGL(k_pow_table, dim3(grid), dim3(256), 0, stream, out, base, mult, log_stride,
                       count, P);
    return hipGetLastError();
}

}  // namespace hodor
