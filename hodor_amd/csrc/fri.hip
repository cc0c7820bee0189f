// fri.hip — K7: the FRI folding rounds on device (round table + fold, and the fused tail of small rounds).
//
// Replaces the per-round loop body of NaiveFriIop::proof_from_lde_by_values
// (/root/reference/src/fri/fri_on_values.rs:70-104):
//     next[i] = ((f[i] + f[i+half]) + beta * (f[i] - f[i+half]) * w^-(i*stride)) * 2^-1
// The reference tabulates all n/2 powers of w^-1 (:24-40, n/2 x 32 B streamed from memory every
// round) and spends three multiplications per output.  Here:
//   * w^-e comes from the two-level table of the initial domain's w^-1 (L2-resident);
//   * beta/2 is folded into the table's `hi` half once per round (k_fri_round_table, a few thousand
//     products), so an output costs one product to combine lo*hi' (none when the low exponent bits
//     are zero, i.e. in every round >= lo_bits) and one to apply it;
//   * (f[i] + f[i+half]) / 2 is an exact halving, not a multiplication;
//   * arithmetic is the carry-free 9 x 29-bit form (fr9.cuh); beta is read from device memory, so
//     the 23-round chain at 2^26 never synchronises with the host.
#include "blake2s.cuh"
#include "ntt.cuh"

namespace hodor {

// Start of a round: the previous tree's root -> challenge beta (interpret_hash, K8), then
// hi'[j] = hi[j] * beta / 2 (all R'-form, normalized).  Every workgroup derives beta from the root
// itself (one Montgomery product) instead of waiting for a separate one-lane kernel; workgroup 0 also
// records beta and the root.  `c16` = 16 in R'-form: the product of the R-form challenge with an
// R'-form entry is R-form, i.e. short by 2^5; 2^5 / 2 = 16.
__global__ void __launch_bounds__(256)
k_fri_round_table(const uint4 *nodes, uint4 *chal_out, uint4 *root_out, const uint4 *hi, uint4 *hi_out,
                  uint64_t count, Fr9 c16, Fr r2, uint32_t shave, Fr9Params Q, FrParams P BXPARAM)
{
    __shared__ uint32_t beta_w[8];
    if (threadIdx.x == 0) {
        const uint4 r0 = *BAT(60, nodes, 2, 2), r1 = nodes[3];   // nodes[1] = the root
        const uint32_t d[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        Fr beta = b2s_digest_to_challenge(d, r2, shave, P);
#pragma unroll
        for (int i = 0; i < 8; i++) beta_w[i] = beta.v[i];
        if (blockIdx.x == 0) {
            fr_store(BATS(1, chal_out, 0, 2), beta);
            *BATS(61, root_out, 0, 2) = r0;
            root_out[1] = r1;
        }
    }
    __syncthreads();
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    Fr beta;
#pragma unroll
    for (int i = 0; i < 8; i++) beta.v[i] = beta_w[i];
    Fr9 b16 = fr9_mul(fr9_unpack(beta), c16, Q);           // beta * 16, R-form, normalized, < 2p
    Fr9 h = fr9_mul(b16, fr9_load48(BAT(2, hi, 3 * j, 3)), Q);       // beta * 16 * h_j * 2^256 = (beta h_j / 2) 2^261
    fr9_store48(BATS(3, hi_out, 3 * j, 3), h);
}

__global__ void __launch_bounds__(256)
k_fri_fold(FoldArgs F, Fr9Params Q BXPARAM)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < F.half; i += stride)
        fr_store(BATS(4, F.dst, 2 * i, 2), fri_fold_one(F, i, Q BXPASS));
}

// The coefficient fold of NaiveFriIop::proof_from_lde_through_coefficients
// (/root/reference/src/fri/mod.rs:194-203): next[i] = a[2i] + beta * a[2i+1], beta read from device memory
// (the challenge the previous tree's root gave, R-form) so the chain of rounds never visits the host.
__global__ void __launch_bounds__(256)
k_fri_fold_coeffs(const uint4 *src, uint4 *dst, uint64_t half, const uint4 *chal, FrParams P BXPARAM)
{
    const Fr beta = fr_load(BAT(5, chal, 0, 2));
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += stride)
        fr_store(BATS(8, dst, 2 * i, 2), fr_add(fr_load(BAT(6, src, 4 * i, 2)), fr_mul(fr_load(BAT(7, src, 4 * i + 2, 2)), beta, P), P));
}

// ---------------------------------------------------------------------------------------------
// k_fri_tail: all remaining rounds of the commit loop once a round's output fits one workgroup
// (<= 512 values).  Below that size a round is a chain of ~log2(size) + 3 dependent steps (fold,
// leaf hashes, one compression per tree level, challenge) of a few microseconds each, and running
// it as 4-5 launches per round costs more in launch-to-launch latency than in work.  One workgroup
// walks the rounds back to back: the fold keeps its output in registers for the leaf hash, tree
// levels hand over through LDS, and the challenge of round i reaches round i+1 through the
// challenge array itself (same workgroup, barrier in between).  Everything the multi-launch path
// writes (values, trees with nodes[0] = 0, roots, challenges) is written identically.
// ---------------------------------------------------------------------------------------------
// COMB: the trees are COSET2 trees (merkle.hip): leaf k of a round with h outputs is y[k] || y[k + h/2] — the two
// lanes that computed them drop their values into the leaf's 64-byte message block in LDS — and the tree over
// the h/2 leaves has h/2 heap entries.  Rounds of fewer than 4 outputs do not occur (abi_fri.hip refuses them).
template <bool COMB>
__global__ void __launch_bounds__(FRI_TAIL_THREADS)
k_fri_tail(FriTailArgs A, Fr9 c16, Fr r2, B2Mid mid, Fr9Params Q, FrParams P BXPARAM)
{
    constexpr uint32_t QUADS = FRI_TAIL_THREADS / 4;       // hashes per pass in quad-lane mode
    __shared__ uint4 buf_a[2 * FRI_TAIL_THREADS];          // leaf hashes, then every other level
    __shared__ uint4 buf_b[FRI_TAIL_THREADS];
    __shared__ uint4 buf_m[4 * QUADS];                     // leaf message blocks (value | zeros) for quad mode
    const uint32_t tid = threadIdx.x, quad = tid >> 2, j = tid & 3;
    const uint64_t lo_mask = (1ull << A.lo_bits) - 1;
    B2Quad bq;
    b2q_init(bq, mid, j);
    const uint4 *src = A.src;
    for (uint32_t k = 0; k < A.rounds; k++) {
        const uint32_t h = A.half0 >> k, gi = A.first_round + k;
        uint4 *dst = A.values[k], *nodes = A.nodes[k];
        const uint32_t leaves = COMB ? h >> 1 : h;
        const bool quad_leafs = leaves <= QUADS;
        if (tid < h) {
            // fold (fri_on_values.rs:77-100), same arithmetic as k_fri_round_table + k_fri_fold
            Fr9 b16 = fr9_mul(fr9_unpack(fr_load(BAT(9, A.chal, 2 * gi, 2))), c16, Q);
            Fr9 a = fr9_unpack(fr_load(BAT(10, src, 2 * tid, 2))), b = fr9_unpack(fr_load(BAT(11, src, 2 * (tid + h), 2)));
            uint64_t e = (uint64_t)tid << gi;
            Fr9 tw = fr9_load48(BAT(12, A.hi, 3 * (e >> A.lo_bits), 3));
            if (e & lo_mask) tw = fr9_mul(tw, fr9_load48(BAT(13, A.lo, 3 * (e & lo_mask), 3)), Q);
            tw = fr9_mul(b16, tw, Q);                            // w^-e * beta / 2, R'-form
            Fr9 odd = fr9_mul(fr9_sub(a, b, Q), tw, Q);
            Fr9 even = fr9_halve(fr9_add(a, b), Q);
            Fr y = fr9_to_canonical(fr9_add(even, odd), Q);
            fr_store(BATS(14, dst, 2 * tid, 2), y);
            const uint4 y0 = make_uint4(y.v[0], y.v[1], y.v[2], y.v[3]), y1 = make_uint4(y.v[4], y.v[5], y.v[6], y.v[7]);
            if (COMB) {
                // message block of leaf tid mod h/2, lower or upper 32 bytes; the blocks of a round too wide for
                // the quad-lane hash (256 leaves) are staged in buf_a, whose hashes then go to buf_b
                uint4 *blk = (quad_leafs ? buf_m : buf_a) + 4 * (tid & (leaves - 1)) + 2 * (tid >= leaves ? 1 : 0);
                blk[0] = y0; blk[1] = y1;
            } else if (quad_leafs) {
                buf_m[4 * tid] = y0; buf_m[4 * tid + 1] = y1;
                buf_m[4 * tid + 2] = make_uint4(0, 0, 0, 0); buf_m[4 * tid + 3] = make_uint4(0, 0, 0, 0);
            } else {
                uint32_t out[8];
                b2s_leaf(mid, y0, y1, out);
                buf_a[2 * tid] = make_uint4(out[0], out[1], out[2], out[3]);
                buf_a[2 * tid + 1] = make_uint4(out[4], out[5], out[6], out[7]);
            }
        }
        __syncthreads();
        if (quad_leafs) {
            if (quad < leaves) {
                uint32_t lo, hi;
                b2q_compress(bq, reinterpret_cast<const uint32_t *>(buf_m + 4 * quad), COMB ? B2Q_PAIR : B2Q_LEAF, lo, hi);
                uint32_t *o = reinterpret_cast<uint32_t *>(buf_a + 2 * quad);
                o[j] = lo; o[4 + j] = hi;
            }
            __syncthreads();
        } else if (COMB) {
            if (tid < leaves) {
                uint32_t out[8];
                b2s_pair(mid, buf_a[4 * tid], buf_a[4 * tid + 1], buf_a[4 * tid + 2], buf_a[4 * tid + 3], out);
                buf_b[2 * tid] = make_uint4(out[0], out[1], out[2], out[3]);
                buf_b[2 * tid + 1] = make_uint4(out[4], out[5], out[6], out[7]);
            }
            __syncthreads();
        }
        uint4 *s = (COMB && !quad_leafs) ? buf_b : buf_a, *d = (COMB && !quad_leafs) ? buf_a : buf_b;
        for (uint32_t w = leaves >> 1; w >= 1; w >>= 1) {        // level of width w at nodes[w .. 2w)
            if (w <= QUADS) {
                if (quad < w) {
                    uint32_t lo, hi;
                    b2q_compress(bq, reinterpret_cast<const uint32_t *>(s + 4 * quad), B2Q_NODE, lo, hi);
                    uint32_t *o = reinterpret_cast<uint32_t *>(d + 2 * quad);
                    o[j] = lo; o[4 + j] = hi;
                    uint32_t *g = reinterpret_cast<uint32_t *>(BATS(62, nodes, 2 * (w + quad), 2));
                    g[j] = lo; g[4 + j] = hi;
                }
            } else if (tid < w) {
                uint4 a0 = s[4 * tid], a1 = s[4 * tid + 1], b0 = s[4 * tid + 2], b1 = s[4 * tid + 3];
                uint32_t l[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                uint32_t r[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                uint32_t out[8];
                b2s_node(mid, l, r, out);
                uint4 o0 = make_uint4(out[0], out[1], out[2], out[3]), o1 = make_uint4(out[4], out[5], out[6], out[7]);
                *BATS(63, nodes, 2 * (w + tid), 2) = o0; nodes[2 * (w + tid) + 1] = o1;
                d[2 * tid] = o0; d[2 * tid + 1] = o1;
            }
            __syncthreads();
            uint4 *t = s; s = d; d = t;
        }
        if (tid == 0) {                                          // s[0..1] = the root: challenge of the next round
            uint4 o0 = s[0], o1 = s[1];
            uint32_t out[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
            *BATS(64, nodes, 0, 2) = make_uint4(0, 0, 0, 0);
            nodes[1] = make_uint4(0, 0, 0, 0);
            *BATS(65, A.roots, 2 * (gi + 1), 2) = o0; A.roots[2 * (gi + 1) + 1] = o1;
            fr_store(BATS(15, A.chal, 2 * (gi + 1), 2), b2s_digest_to_challenge(out, r2, A.shave, P));
        }
        __syncthreads();
        src = dst;
    }
}

hipError_t fri_tail_launch(hipStream_t s, const FriTailArgs &A, const Fr9 &c16, const Fr &r2, const B2Mid &mid,
                           const Fr9Params &Q, const FrParams &P, bool comb)
{
    BX_BEGIN(bx, KID_FRI_TAIL);
    BX_ADD(bx, A.src, 2ull * A.half0 * 32);
    for (uint32_t k = 0; k < A.rounds; k++) {
        BX_ADD(bx, A.values[k], (uint64_t)(A.half0 >> k) * 32);
        BX_ADD(bx, A.nodes[k], (uint64_t)((A.half0 >> k) >> (comb ? 1 : 0)) * 32);
    }
    BX_ADD(bx, A.chal, (uint64_t)(A.first_round + A.rounds + 1) * 32);
    BX_ADD(bx, A.roots, (uint64_t)(A.first_round + A.rounds + 1) * 32);
#ifdef HODOR_BOUNDS
    bx.add(A.lo, A.lo_bytes).add(A.hi, A.hi_bytes);
#endif
    if (comb) hipLaunchKernelGGL(k_fri_tail<true>, dim3(1), dim3(FRI_TAIL_THREADS), 0, s, A, c16, r2, mid, Q, P BXARG(bx));
    else hipLaunchKernelGGL(k_fri_tail<false>, dim3(1), dim3(FRI_TAIL_THREADS), 0, s, A, c16, r2, mid, Q, P BXARG(bx));
    return hipGetLastError();
}

hipError_t fri_round_table_launch(hipStream_t s, const uint4 *nodes, uint4 *chal_out, uint4 *root_out,
                                  const uint4 *hi, uint4 *hi_out, uint64_t count, const Fr9 &c16, const Fr &r2,
                                  uint32_t shave, const Fr9Params &Q, const FrParams &P)
{
    BX_BEGIN(bx, KID_FRI_ROUND_TABLE);
    BX_ADD(bx, nodes, 64);
    BX_ADD(bx, chal_out, 32);
    BX_ADD(bx, root_out, 32);
    BX_ADD(bx, hi, count * 48);
    BX_ADD(bx, hi_out, count * 48);
    hipLaunchKernelGGL(k_fri_round_table, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, nodes, chal_out,
                       root_out, hi, hi_out, count, c16, r2, shave, Q, P BXARG(bx));
    return hipGetLastError();
}

hipError_t fri_fold_coeffs_launch(hipStream_t s, const uint4 *src, uint4 *dst, uint64_t half, const uint4 *chal,
                                  const FrParams &P)
{
    uint64_t blocks = (half + 255) / 256;
    unsigned grid = (unsigned)(blocks < 4096 ? (blocks ? blocks : 1) : 4096);
    BX_BEGIN(bx, KID_FRI_FOLD_COEFFS);
    BX_ADD(bx, src, 2 * half * 32);
    BX_ADD(bx, dst, half * 32);
    BX_ADD(bx, chal, 32);
    hipLaunchKernelGGL(k_fri_fold_coeffs, dim3(grid), dim3(256), 0, s, src, dst, half, chal, P BXARG(bx));
    return hipGetLastError();
}

hipError_t fri_fold_launch(hipStream_t s, const FoldArgs &F, const Fr9Params &Q)
{
    uint64_t blocks = (F.half + 255) / 256;
    unsigned grid = (unsigned)(blocks < 4096 ? (blocks ? blocks : 1) : 4096);
    BX_BEGIN(bx, KID_FRI_FOLD);
    BX_ADD(bx, F.src, 2 * F.half * 32);
    BX_ADD(bx, F.dst, F.half * 32);
#ifdef HODOR_BOUNDS
    bx.add(F.lo, F.lo_bytes).add(F.hi_beta, F.hi_bytes);
#endif
    hipLaunchKernelGGL(k_fri_fold, dim3(grid), dim3(256), 0, s, F, Q BXARG(bx));
    return hipGetLastError();
}

}  // namespace hodor
