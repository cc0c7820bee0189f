// fri.hip — K7: one FRI folding round on device.
//
// Replaces the per-round loop body of NaiveFriIop::proof_from_lde_by_values
// (/root/reference/src/fri/fri_on_values.rs:70-104):
//     next[i] = ((f[i] + f[i+half]) + beta * (f[i] - f[i+half]) * w^-(i*stride)) * 2^-1
// The reference tabulates all n/2 powers of w^-1 (:24-40, n/2 x 32 B streamed from memory every
// round); here they come from the two-level table of the initial domain's w^-1 (L2-resident) and
// the final halving is an exact shift (add p if odd, >> 1) instead of a multiplication by 2^-1.
// beta is read from device memory, so the round chain never synchronises with the host.
#include "ntt.cuh"

namespace hodor {

__device__ __forceinline__ Fr tl_pow(const TwoLevel &t, uint64_t e, const FrParams &P)
{
    uint64_t lo_i = e & ((1ull << t.lo_bits) - 1), hi_i = e >> t.lo_bits;
    Fr h = fr_load(t.hi + 2 * hi_i);
    if (lo_i == 0) return h;
    return fr_mul(h, fr_load(t.lo + 2 * lo_i), P);
}

__global__ void __launch_bounds__(256)
k_fri_fold(const uint4 *src, uint4 *dst, uint64_t half, TwoLevel winv, uint32_t log_stride,
           const uint4 *challenge, FrParams P)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    Fr beta = fr_load(challenge);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += stride) {
        Fr a = fr_load(src + 2 * i), b = fr_load(src + 2 * (i + half));
        Fr even = fr_add(a, b, P);
        Fr odd = fr_sub(a, b, P);
        if (i != 0) odd = fr_mul(odd, tl_pow(winv, i << log_stride, P), P);
        Fr t = fr_add(fr_mul(odd, beta, P), even, P);
        fr_store(dst + 2 * i, fr_halve(t, P));
    }
}

hipError_t fri_fold_launch(hipStream_t s, const uint4 *src, uint4 *dst, uint64_t half,
                           const TwoLevel &winv, uint32_t log_stride, const uint4 *challenge,
                           const FrParams &P)
{
    uint64_t blocks = (half + 255) / 256;
    unsigned grid = (unsigned)(blocks < 4096 ? (blocks ? blocks : 1) : 4096);
    hipLaunchKernelGGL(k_fri_fold, dim3(grid), dim3(256), 0, s, src, dst, half, winv, log_stride,
                       challenge, P);
    return hipGetLastError();
}

}  // namespace hodor
