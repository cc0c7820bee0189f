// pointwise.hip — K4/K9: streaming element-wise kernels over `&mut [F]` buffers.
//
//   k_distribute_powers   a[i] *= g^i          /root/reference/src/fft/mod.rs:110-123
//   k_scale               a[i] *= s            /root/reference/src/polynomials/mod.rs:60-72
//   k_binary              a[i] (+,-,*)= b[i]   /root/reference/src/polynomials/mod.rs:817-887
//
// HBM-streaming kernels: grid-stride over 32-byte elements, one element per lane per iteration so a
// wave touches 2 KiB of contiguous memory; powers are a per-thread running product (one
// square-and-multiply at entry, then one multiply by g^stride per iteration).
#include "ntt.cuh"

namespace hodor {

__global__ void __launch_bounds__(256)
k_distribute_powers(uint4 *a, uint64_t n, Fr g, FrParams P)
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

static unsigned stream_grid(uint64_t n)
{
    uint64_t blocks = (n + 255) / 256;
    return (unsigned)(blocks < 2048 ? (blocks ? blocks : 1) : 2048);
}

hipError_t distribute_powers_launch(hipStream_t s, uint4 *a, uint64_t n, const Fr &g, const FrParams &P)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_distribute_powers, dim3(stream_grid(n)), dim3(256), 0, s, a, n, g, P);
    return hipGetLastError();
}

hipError_t scale_launch(hipStream_t s, uint4 *a, uint64_t n, const Fr &f, const FrParams &P)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scale, dim3(stream_grid(n)), dim3(256), 0, s, a, n, f, P);
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

}  // namespace hodor
