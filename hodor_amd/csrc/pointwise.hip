// pointwise.hip — K4/K9: streaming element-wise kernels over `&mut [F]` buffers.
//
//   k_distribute_powers   a[i] *= g^i          /root/reference/src/fft/mod.rs:110-123
//   k_scale               a[i] *= s            /root/reference/src/polynomials/mod.rs:60-72
//   k_binary              a[i] (+,-,*)= b[i]   /root/reference/src/polynomials/mod.rs:817-887
//   k_add_scaled / k_unary / k_batchinv_* / k_evaluate_at : the rest of the value-form API that sits
//                         either side of every LDE in ALI (src/polynomials/mod.rs:657-711, 744-954)
//
// HBM-streaming kernels: grid-stride over 32-byte elements, one element per lane per iteration so a
// wave touches 2 KiB of contiguous memory; powers come from the cached two-level tables (abi.hip).
#include "ntt.cuh"

namespace hodor {

// small inputs: per-thread running product (one square-and-multiply at entry, then one multiply by
// g^stride per iteration) — no table to build for a generator that may never come back
__global__ void __launch_bounds__(256)
k_distribute_powers_small(uint4 *a, uint64_t n, Fr g, FrParams P)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr u = fr_pow(g, i, P);
    Fr step = fr_pow(g, stride, P);
    for (; i < n; i += stride) {
        Fr x = fr_load(a + 2 * i);
        fr_store(a + 2 * i, fr_mul(x, u, P));
        u = fr_mul(u, step, P);
    }
}

// g^i from the cached two-level table of g (lo[i & mask] * hi[i >> bits], L2-resident, R'-form): two
// products per element in the 9 x 29 arithmetic, no per-thread square-and-multiply and no dependent
// running product.
__global__ void __launch_bounds__(256)
k_distribute_powers(uint4 *a, uint64_t n, TwoLevel t, Fr9Params Q)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t lo_mask = (1ull << t.lo_bits) - 1;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        Fr9 w = fr9_load48(t.hi + 3 * (i >> t.lo_bits));
        if (i & lo_mask) w = fr9_mul(w, fr9_load48(t.lo + 3 * (i & lo_mask)), Q);
        Fr9 x = fr9_mul(fr9_unpack(fr_load(a + 2 * i)), w, Q);
        fr_store(a + 2 * i, fr9_to_canonical<true>(x, Q));
    }
}

// Synthetic input, index-addressable: element i is the first of 16 candidates
// (four SplitMix64 outputs each, top limb masked to the field's bit length) below p, converted to
// Montgomery form.  Bit-identical to oracle/hodor_oracle.c:o_gen_elements, which documents the stream.
__device__ __forceinline__ uint64_t splitmix64_out(uint64_t seed, uint64_t m)
{
    uint64_t z = seed + (m + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void __launch_bounds__(256)
k_gen_elements(uint4 *out, uint64_t first, uint64_t count, uint64_t seed, uint64_t top_mask, Fr r2, FrParams P)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < count; r += stride) {
        const uint64_t i = first + r;
        Fr x;
        bool ok = false;
        for (uint64_t t = 0; t < 16 && !ok; t++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint64_t w = splitmix64_out(seed, 4 * (16 * i + t) + k);
                if (k == 3) w &= top_mask;
                x.v[2 * k] = (uint32_t)w;
                x.v[2 * k + 1] = (uint32_t)(w >> 32);
            }
            ok = false;   // x < p ?
#pragma unroll
            for (int k = 7; k >= 0; k--) {
                if (x.v[k] != P.p[k]) { ok = x.v[k] < P.p[k]; break; }
            }
        }
        if (!ok) { x.v[6] = 0; x.v[7] = 0; }
        fr_store(out + 2 * r, fr_mul(x, r2, P));
    }
}

__global__ void __launch_bounds__(256)
k_scale(uint4 *a, uint64_t n, Fr s, FrParams P)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        fr_store(a + 2 * i, fr_mul(fr_load(a + 2 * i), s, P));
}

// op: 0 add, 1 sub, 2 mul
__global__ void __launch_bounds__(256)
k_binary(uint4 *a, const uint4 *b, uint64_t n, int op, FrParams P)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        Fr x = fr_load(a + 2 * i), y = fr_load(b + 2 * i), r;
        if (op == 0) r = fr_add(x, y, P);
        else if (op == 1) r = fr_sub(x, y, P);
        else r = fr_mul(x, y, P);
        fr_store(a + 2 * i, r);
    }
}

// six-step twiddle: a[r][c] *= w^((row0 + r) * c mod 2^log_order) (* scale), a is rows x cols row-major.
// w powers from the two-level table of w (order 2^log_order).
__global__ void __launch_bounds__(256)
k_twiddle_mul(uint4 *a, uint64_t rows, uint64_t cols, uint64_t row0, TwoLevel t, uint32_t log_order,
              Fr scale, uint32_t has_scale, FrParams P)
{
    const uint64_t total = rows * cols;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t emask = (1ull << log_order) - 1;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        uint64_t r = i / cols, c = i - r * cols;
        uint64_t e = ((row0 + r) * c) & emask;
        Fr x = fr_load(a + 2 * i);
        if (e != 0) {
            uint64_t lo_i = e & ((1ull << t.lo_bits) - 1), hi_i = e >> t.lo_bits;
            Fr w = fr_load(t.hi + 2 * hi_i);
            if (lo_i) w = fr_mul(w, fr_load(t.lo + 2 * lo_i), P);
            x = fr_mul(x, w, P);
        }
        if (has_scale) x = fr_mul(x, scale, P);
        fr_store(a + 2 * i, x);
    }
}

// a[i] += s * b[i]   (Polynomial<Coefficients>::add_assign_scaled, src/polynomials/mod.rs:657-671)
__global__ void __launch_bounds__(256)
k_add_scaled(uint4 *a, const uint4 *b, uint64_t n, Fr s, FrParams P)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        fr_store(a + 2 * i, fr_add(fr_load(a + 2 * i), fr_mul(fr_load(b + 2 * i), s, P), P));
}

// unary ops of Polynomial<F, Values> (src/polynomials/mod.rs:60-83, 744-771, 817-841):
// 0 negate, 1 square, 2 pow(e), 3 scale(c), 4 add_constant(c), 5 sub_constant(c)
__global__ void __launch_bounds__(256)
k_unary(uint4 *a, uint64_t n, int op, Fr c, uint64_t e, FrParams P)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        Fr x = fr_load(a + 2 * i), r;
        switch (op) {
        case 0: r = fr_is_zero(x) ? x : fr_neg(x, P); break;
        case 1: r = fr_sqr(x, P); break;
        case 2: r = fr_pow(x, e, P); break;
        case 3: r = fr_mul(x, c, P); break;
        case 4: r = fr_add(x, c, P); break;
        default: r = fr_sub(x, c, P); break;
        }
        fr_store(a + 2 * i, r);
    }
}

// Polynomial<F, Values>::batch_inversion (src/polynomials/mod.rs:889-954).  Montgomery's trick, but
// hierarchical so that every level runs with as many threads as the chip holds instead of one long
// dependent product chain per CPU-style worker:
//   forward (k_batchinv_forward): thread t of T owns the strided subsequence a[t], a[t+T], ...
//     (coalesced across lanes; 8 elements), writes the running prefix product of each element
//     to scratch and the product of the whole subsequence to prod[t]; flags zero elements — the
//     reference errors out before touching the data (:909), and so does the caller here;
//   the T subsequence products are inverted by the same procedure (T/8 threads, ...) until at most
//     16 are left, which the host inverts (it has to look at the zero flag at that point anyway);
//   backward (k_batchinv_backward): a[i] = inv * prefix[i], inv *= old a[i], walking the subsequence
//     from its end.
// Three products per element plus ~1/8 for the upper levels; 160 bytes of traffic per element.
__global__ void __launch_bounds__(256)
k_batchinv_forward(const uint4 *a, uint64_t n, uint64_t T, uint4 *prefix, uint4 *prod, uint32_t *zero_flag,
                   FrParams P)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    Fr run = fr_one(P);
    bool zero = false;
    for (uint64_t i = t; i < n; i += T) {           // prefix[i] = product of this thread's elements before i
        Fr x = fr_load(a + 2 * i);
        zero |= fr_is_zero(x);
        fr_store(prefix + 2 * i, run);
        run = fr_mul(run, x, P);
    }
    fr_store(prod + 2 * t, run);
    if (zero && zero_flag) atomicOr(zero_flag, 1u);
}

__global__ void __launch_bounds__(256)
k_batchinv_backward(uint4 *a, uint64_t n, uint64_t T, const uint4 *prefix, const uint4 *prod_inv, FrParams P)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T || t >= n) return;
    Fr inv = fr_load(prod_inv + 2 * t);             // (product of the whole subsequence)^-1
    const uint64_t last = t + ((n - 1 - t) / T) * T;
    for (uint64_t i = last;; i -= T) {
        Fr x = fr_load(a + 2 * i);
        fr_store(a + 2 * i, fr_mul(inv, fr_load(prefix + 2 * i), P));
        inv = fr_mul(inv, x, P);
        if (i == t) break;
    }
}

// Polynomial<F, Coefficients>::evaluate_at (src/polynomials/mod.rs:685-711): sum a[i] g^i.
// Per-thread strided partial sums, then a workgroup tree in LDS; one partial per workgroup goes to
// `partials`, the last workgroup to finish (ticket) folds them into out[0].
__device__ __forceinline__ void evaluate_reduce(Fr acc, uint4 *partials, uint32_t *ticket, uint4 *out,
                                                const FrParams &P)
{
    __shared__ uint4 red[2 * 256];
    __shared__ bool is_last;
    fr_store(red + 2 * threadIdx.x, acc);
    __syncthreads();
    for (uint32_t w = 128; w >= 1; w >>= 1) {
        if (threadIdx.x < w)
            fr_store(red + 2 * threadIdx.x,
                     fr_add(fr_load(red + 2 * threadIdx.x), fr_load(red + 2 * (threadIdx.x + w)), P));
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        fr_store(partials + 2 * blockIdx.x, fr_load(red));
        __threadfence();                                   // publish the partial before taking a ticket
        is_last = (atomicAdd(ticket, 1u) == gridDim.x - 1);
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    Fr s = fr_zero();
    for (uint32_t b = threadIdx.x; b < gridDim.x; b += 256) {
        const volatile uint32_t *q = reinterpret_cast<const volatile uint32_t *>(partials + 2 * b);
        Fr x;
#pragma unroll
        for (int k = 0; k < 8; k++) x.v[k] = q[k];         // bypass a possibly stale L1 line
        s = fr_add(s, x, P);
    }
    fr_store(red + 2 * threadIdx.x, s);
    __syncthreads();
    for (uint32_t w = 128; w >= 1; w >>= 1) {
        if (threadIdx.x < w)
            fr_store(red + 2 * threadIdx.x,
                     fr_add(fr_load(red + 2 * threadIdx.x), fr_load(red + 2 * (threadIdx.x + w)), P));
        __syncthreads();
    }
    if (threadIdx.x == 0) fr_store(out, fr_load(red));
}

// small inputs: running power per thread (one square-and-multiply at entry)
__global__ void __launch_bounds__(256)
k_evaluate_at(const uint4 *a, uint64_t n, Fr g, uint4 *partials, uint32_t *ticket, uint4 *out, FrParams P)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fr acc = fr_zero();
    if (i < n) {
        Fr u = fr_pow(g, i, P), step = fr_pow(g, stride, P);
        for (; i < n; i += stride) {
            acc = fr_add(acc, fr_mul(fr_load(a + 2 * i), u, P), P);
            u = fr_mul(u, step, P);
        }
    }
    evaluate_reduce(acc, partials, ticket, out, P);
}

// large inputs: g^i = lo[i & mask] * hi[i >> bits] from the cached two-level table.  The thread count T
// is a multiple of 2^bits, so a thread's elements i0, i0 + T, ... share their `lo` factor: one product
// per element with the `hi` entry, lazily accumulated (9 x 29 limbs, at most 64 terms), and a single
// product with lo[i0 & mask] at the end.
__global__ void __launch_bounds__(256)
k_evaluate_at_table(const uint4 *a, uint64_t n, TwoLevel t, uint4 *partials, uint32_t *ticket, uint4 *out,
                    Fr9Params Q, FrParams P)
{
    const uint64_t T = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fr9 acc;
#pragma unroll
    for (int k = 0; k < 9; k++) acc.v[k] = 0;
    uint32_t pending = 0;
    for (uint64_t i = i0; i < n; i += T) {
        Fr9 term = fr9_mul(fr9_unpack(fr_load(a + 2 * i)), fr9_load48(t.hi + 3 * (i >> t.lo_bits)), Q);
        acc = fr9_add(acc, term);                           // terms are normalized and < 2p
        if (++pending == 4) { fr9_normalize(acc); pending = 0; }
    }
    fr9_normalize(acc);
    acc = fr9_mul(acc, fr9_load48(t.lo + 3 * (i0 & ((1ull << t.lo_bits) - 1))), Q);
    evaluate_reduce(fr9_to_canonical<true>(acc, Q), partials, ticket, out, P);
}

static unsigned stream_grid(uint64_t n)
{
    uint64_t blocks = (n + 255) / 256;
    return (unsigned)(blocks < 2048 ? (blocks ? blocks : 1) : 2048);
}

hipError_t distribute_powers_small_launch(hipStream_t s, uint4 *a, uint64_t n, const Fr &g, const FrParams &P)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_distribute_powers_small, dim3(stream_grid(n)), dim3(256), 0, s, a, n, g, P);
    return hipGetLastError();
}

hipError_t distribute_powers_launch(hipStream_t s, uint4 *a, uint64_t n, const TwoLevel &t, const Fr9Params &Q)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_distribute_powers, dim3(stream_grid(n)), dim3(256), 0, s, a, n, t, Q);
    return hipGetLastError();
}

hipError_t scale_launch(hipStream_t s, uint4 *a, uint64_t n, const Fr &f, const FrParams &P)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scale, dim3(stream_grid(n)), dim3(256), 0, s, a, n, f, P);
    return hipGetLastError();
}

hipError_t add_scaled_launch(hipStream_t s, uint4 *a, const uint4 *b, uint64_t n, const Fr &f, const FrParams &P)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_add_scaled, dim3(stream_grid(n)), dim3(256), 0, s, a, b, n, f, P);
    return hipGetLastError();
}

hipError_t unary_launch(hipStream_t s, uint4 *a, uint64_t n, int op, const Fr &c, uint64_t e, const FrParams &P)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_unary, dim3(stream_grid(n)), dim3(256), 0, s, a, n, op, c, e, P);
    return hipGetLastError();
}

// batch inversion launchers (the level recursion lives in abi.hip, which owns the scratch)
hipError_t batchinv_forward_launch(hipStream_t s, const uint4 *a, uint64_t n, uint64_t T, uint4 *prefix, uint4 *prod,
                                   uint32_t *zero_flag, const FrParams &P)
{
    hipLaunchKernelGGL(k_batchinv_forward, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, s, a, n, T, prefix, prod,
                       zero_flag, P);
    return hipGetLastError();
}

hipError_t batchinv_backward_launch(hipStream_t s, uint4 *a, uint64_t n, uint64_t T, const uint4 *prefix,
                                    const uint4 *prod_inv, const FrParams &P)
{
    hipLaunchKernelGGL(k_batchinv_backward, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, s, a, n, T, prefix,
                       prod_inv, P);
    return hipGetLastError();
}

// one partial per workgroup; `n >> 5` threads (a multiple of 2^lo_bits for n >= 2^16), <= 32 terms each
unsigned evaluate_at_table_blocks(uint32_t log_n) { return 1u << (log_n - 5 - 8); }
hipError_t evaluate_at_table_launch(hipStream_t s, const uint4 *a, uint64_t n, uint32_t log_n, const TwoLevel &t,
                                    uint4 *partials, uint32_t *ticket, uint4 *out, const Fr9Params &Q,
                                    const FrParams &P)
{
    hipLaunchKernelGGL(k_evaluate_at_table, dim3(evaluate_at_table_blocks(log_n)), dim3(256), 0, s, a, n, t, partials,
                       ticket, out, Q, P);
    return hipGetLastError();
}

// `work` must hold 32 * 256 + 4 bytes (partials + ticket, ticket zeroed by the caller)
hipError_t evaluate_at_launch(hipStream_t s, const uint4 *a, uint64_t n, const Fr &g, uint4 *partials,
                              uint32_t *ticket, uint4 *out, const FrParams &P)
{
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 256) blocks = 256;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(k_evaluate_at, dim3((unsigned)blocks), dim3(256), 0, s, a, n, g, partials, ticket, out, P);
    return hipGetLastError();
}

hipError_t twiddle_mul_launch(hipStream_t s, uint4 *a, uint64_t rows, uint64_t cols, uint64_t row0,
                              const TwoLevel &t, uint32_t log_order, const Fr *scale, const FrParams &P)
{
    if (rows * cols == 0) return hipSuccess;
    Fr sc = {};
    if (scale) sc = *scale;
    hipLaunchKernelGGL(k_twiddle_mul, dim3(stream_grid(rows * cols)), dim3(256), 0, s, a, rows, cols, row0, t,
                       log_order, sc, scale ? 1u : 0u, P);
    return hipGetLastError();
}

hipError_t binary_launch(hipStream_t s, uint4 *a, const uint4 *b, uint64_t n, int op, const FrParams &P)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_binary, dim3(stream_grid(n)), dim3(256), 0, s, a, b, n, op, P);
    return hipGetLastError();
}

hipError_t gen_elements_launch(hipStream_t stream, uint4 *out, uint64_t first, uint64_t count, uint64_t seed,
                               uint64_t top_mask, const Fr &r2, const FrParams &P)
{
    if (count == 0) return hipSuccess;
    uint64_t blocks = (count + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_gen_elements, dim3((unsigned)blocks), dim3(256), 0, stream, out, first, count, seed,
                       top_mask, r2, P);
    return hipGetLastError();
}

}  // namespace hodor
