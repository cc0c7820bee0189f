// merkle.hip — K6/K8: keyed BLAKE2s leaf hashing + Merkle tree build, root -> field challenge.
//
// Replaces (paths relative to /root/reference):
//   Blake2sLeafEncoder::encode_leaf / Blake2sTreeHasher::{hash_leaf,hash_node}
//                                                   src/iop/blake2s_trivial_iop.rs:36-42, 81-104
//   Blake2sIopTree::create                          src/iop/blake2s_trivial_iop.rs:131-219
//   interpret_hash / get_challenge_scalar_from_root src/iop/blake2s_trivial_iop.rs:48-60, 226-234
//
// BLAKE2s-256, key "Squeamish Ossifrage", personal "Shaftoe" (:8-16).  The keyed first block is the
// same for every hash, so its chaining value (the "midstate") is computed once on the host and passed
// as a kernel argument: each leaf/node hash is ONE compression here instead of the CPU path's two.
// Leaf bytes are the 32 bytes of the Montgomery limbs as they sit in memory (little-endian host).
//
// Tree layout = the reference's heap array: nodes[1] root, level l at nodes[2^l .. 2^(l+1)),
// nodes[0] unused (zeroed), leaf hashes are not stored.
//
// 32-bit ARX integer work: no MFMA.  One hash per lane, all 16 state words + 16 message words in
// VGPRs, sigma schedule resolved at compile time.
#include <cstdlib>

#include "fr.cuh"

namespace hodor {

__device__ __forceinline__ uint32_t rotr32(uint32_t x, int n)
{
    return __builtin_rotateright32(x, n);
}

#define B2S_G(a, b, c, d, x, y)                                   \
    do {                                                          \
        a = a + b + (x); d = rotr32(d ^ a, 16);                   \
        c = c + d;       b = rotr32(b ^ c, 12);                   \
        a = a + b + (y); d = rotr32(d ^ a, 8);                    \
        c = c + d;       b = rotr32(b ^ c, 7);                    \
    } while (0)

#define B2S_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
    do {                                                                                  \
        B2S_G(v0, v4, v8, v12, m[s0], m[s1]);                                             \
        B2S_G(v1, v5, v9, v13, m[s2], m[s3]);                                             \
        B2S_G(v2, v6, v10, v14, m[s4], m[s5]);                                            \
        B2S_G(v3, v7, v11, v15, m[s6], m[s7]);                                            \
        B2S_G(v0, v5, v10, v15, m[s8], m[s9]);                                            \
        B2S_G(v1, v6, v11, v12, m[s10], m[s11]);                                          \
        B2S_G(v2, v7, v8, v13, m[s12], m[s13]);                                           \
        B2S_G(v3, v4, v9, v14, m[s14], m[s15]);                                           \
    } while (0)

// One final-block compression on top of the key-block midstate.  `t` = total bytes incl. the
// 64-byte key block (96 for a leaf, 128 for a node).
__device__ __forceinline__ void b2s_final(const B2Mid &mid, const uint32_t m[16], uint32_t t,
                                          uint32_t out[8])
{
    uint32_t v0 = mid.h[0], v1 = mid.h[1], v2 = mid.h[2], v3 = mid.h[3];
    uint32_t v4 = mid.h[4], v5 = mid.h[5], v6 = mid.h[6], v7 = mid.h[7];
    uint32_t v8 = 0x6A09E667u, v9 = 0xBB67AE85u, v10 = 0x3C6EF372u, v11 = 0xA54FF53Au;
    uint32_t v12 = 0x510E527Fu ^ t, v13 = 0x9B05688Cu, v14 = ~0x1F83D9ABu, v15 = 0x5BE0CD19u;
    B2S_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    B2S_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3);
    B2S_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4);
    B2S_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8);
    B2S_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13);
    B2S_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9);
    B2S_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11);
    B2S_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10);
    B2S_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5);
    B2S_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0);
    out[0] = mid.h[0] ^ v0 ^ v8;  out[1] = mid.h[1] ^ v1 ^ v9;
    out[2] = mid.h[2] ^ v2 ^ v10; out[3] = mid.h[3] ^ v3 ^ v11;
    out[4] = mid.h[4] ^ v4 ^ v12; out[5] = mid.h[5] ^ v5 ^ v13;
    out[6] = mid.h[6] ^ v6 ^ v14; out[7] = mid.h[7] ^ v7 ^ v15;
}

// hash of one 32-byte leaf (message words 8..15 are zero and fold away at compile time)
__device__ __forceinline__ void b2s_leaf(const B2Mid &mid, const uint4 &lo, const uint4 &hi,
                                         uint32_t out[8])
{
    uint32_t m[16] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w, 0, 0, 0, 0, 0, 0, 0, 0};
    b2s_final(mid, m, 96, out);
}

__device__ __forceinline__ void b2s_node(const B2Mid &mid, const uint32_t l[8], const uint32_t r[8],
                                         uint32_t out[8])
{
    uint32_t m[16];
#pragma unroll
    for (int i = 0; i < 8; i++) { m[i] = l[i]; m[8 + i] = r[i]; }
    b2s_final(mid, m, 128, out);
}

// ---------------------------------------------------------------------------------------------
// k_merkle_subtree: one workgroup builds the complete subtree over a chunk of CH = 2^log_ch inputs
// (leaves when LEAF, digests of tree level `m` otherwise): every level of the chunk is written to its
// place in the heap array and handed to the next level through LDS, so each leaf / digest is read
// from HBM once and a 2^25-leaf tree takes 3 launches instead of 17.
//
//   level k of the chunk (k = 1 .. log_ch) has CH >> k nodes at nodes[(m >> k) + chunk*(CH >> k) ..)
//
// Phase 0 pairs two adjacent inputs per lane (64 contiguous bytes; a wave reads 4 KiB contiguous);
// later levels re-map the surviving nodes densely onto the lanes, so the waves stay full until the
// level is narrower than the workgroup.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t MERKLE_LOG_CH = 11;   // 2048 inputs per workgroup: 32 KiB + 16 KiB of LDS

template <bool LEAF>
__global__ void __launch_bounds__(256)
k_merkle_subtree(const uint4 *leafs, uint4 *nodes, uint64_t m, uint32_t log_ch, uint32_t levels, uint64_t n,
                 B2Mid mid)
{
    // blockIdx.y selects one of several independent trees over n leaves each (batched commit)
    leafs += 2 * (uint64_t)blockIdx.y * n;
    nodes += 2 * (uint64_t)blockIdx.y * n;
    __shared__ uint4 buf_a[2 * (1u << (MERKLE_LOG_CH - 1))];   // up to 1024 digests
    __shared__ uint4 buf_b[2 * (1u << (MERKLE_LOG_CH - 2))];   // up to 512 digests
    const uint32_t tid = threadIdx.x;
    const uint32_t ch = 1u << log_ch;
    const uint64_t chunk = blockIdx.x;

    // level 1: pairs of inputs
    const uint4 *in = LEAF ? leafs + 2 * (chunk << log_ch) : nodes + 2 * (m + (chunk << log_ch));
    uint4 *lvl_out = nodes + 2 * ((m >> 1) + chunk * (ch >> 1));
    for (uint32_t p = tid; p < (ch >> 1); p += 256) {
        const uint4 *q = in + 4 * p;
        uint4 a0 = q[0], a1 = q[1], b0 = q[2], b1 = q[3];
        uint32_t l[8], r[8], out[8];
        if (LEAF) {
            b2s_leaf(mid, a0, a1, l);
            b2s_leaf(mid, b0, b1, r);
        } else {
            l[0] = a0.x; l[1] = a0.y; l[2] = a0.z; l[3] = a0.w; l[4] = a1.x; l[5] = a1.y; l[6] = a1.z; l[7] = a1.w;
            r[0] = b0.x; r[1] = b0.y; r[2] = b0.z; r[3] = b0.w; r[4] = b1.x; r[5] = b1.y; r[6] = b1.z; r[7] = b1.w;
        }
        b2s_node(mid, l, r, out);
        uint4 o0 = make_uint4(out[0], out[1], out[2], out[3]), o1 = make_uint4(out[4], out[5], out[6], out[7]);
        lvl_out[2 * p] = o0; lvl_out[2 * p + 1] = o1;
        buf_a[2 * p] = o0; buf_a[2 * p + 1] = o1;
    }
    __syncthreads();

    // levels 2 .. log_ch: ping-pong between the two LDS buffers
    uint4 *src = buf_a, *dst = buf_b;
    for (uint32_t k = 2; k <= levels; k++) {
        const uint32_t w = ch >> k;
        lvl_out = nodes + 2 * ((m >> k) + chunk * w);
        for (uint32_t g = tid; g < w; g += 256) {
            uint4 a0 = src[4 * g], a1 = src[4 * g + 1], b0 = src[4 * g + 2], b1 = src[4 * g + 3];
            uint32_t l[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            uint32_t r[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            uint32_t out[8];
            b2s_node(mid, l, r, out);
            uint4 o0 = make_uint4(out[0], out[1], out[2], out[3]), o1 = make_uint4(out[4], out[5], out[6], out[7]);
            lvl_out[2 * g] = o0; lvl_out[2 * g + 1] = o1;
            dst[2 * g] = o0; dst[2 * g + 1] = o1;
        }
        __syncthreads();
        uint4 *t = src; src = dst; dst = t;
    }
    // the launch that produces the root also clears nodes[0], which the heap layout leaves unused
    if ((m >> levels) == 1 && tid == 0) {
        nodes[0] = make_uint4(0, 0, 0, 0);
        nodes[1] = make_uint4(0, 0, 0, 0);
    }
}

// K8: root digest -> field challenge (interpret_hash): big-endian read, clear the top
// 256 - CAPACITY bits, convert to Montgomery form (multiply by R^2).
// `root_out` (optional) receives a copy of the root digest: the FRI round loop collects its roots there.
__global__ void k_challenge(const uint4 *nodes, uint4 *out, uint4 *root_out, Fr r2, uint32_t shave_bits,
                            FrParams P)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (root_out) {
        root_out[0] = nodes[2];
        root_out[1] = nodes[3];
    }
    const uint32_t *d = reinterpret_cast<const uint32_t *>(nodes + 2);   // nodes[1]
    Fr x;
#pragma unroll
    for (int i = 0; i < 8; i++) x.v[i] = __builtin_bswap32(d[7 - i]);   // byte-reverse 32 bytes
    uint32_t s = shave_bits & 63;   // the reference shifts a 64-bit mask by SHAVE_BITS % 64
    uint64_t mask = ~0ull >> s;
    x.v[7] &= (uint32_t)(mask >> 32);
    x.v[6] &= (uint32_t)mask;
    Fr m = fr_mul(x, r2, P);
    fr_store(out, m);
}

// Query phase on device-resident oracles: IOP::query + IopTree::get_path
// (src/iop/blake2s_trivial_iop.rs:251-279, 324-338).  out[0] = the queried leaf value,
// out[1] = hash of its sibling leaf, out[1 + k] = sibling node at tree level log2(n) - 1 - k.
// `leaf_pair` points at the even leaf of the pair {index & ~1, index | 1}.
__global__ void k_iop_query(const uint4 *leaf_pair, const uint4 *nodes, uint64_t n, uint64_t index,
                            uint4 *out, B2Mid mid)
{
    const uint32_t lane = threadIdx.x;
    uint32_t levels = 0;
    for (uint64_t w = n >> 1; w >= 2; w >>= 1) levels++;   // log2(n) - 1 node levels on the path
    if (lane == 0) {
        const uint4 *me = leaf_pair + 2 * (index & 1), *sib = leaf_pair + 2 * ((index & 1) ^ 1);
        out[0] = me[0];
        out[1] = me[1];
        uint32_t h[8];
        b2s_leaf(mid, sib[0], sib[1], h);
        out[2] = make_uint4(h[0], h[1], h[2], h[3]);
        out[3] = make_uint4(h[4], h[5], h[6], h[7]);
    } else if (lane <= levels) {
        uint32_t k = lane - 1;                       // k-th node level from the bottom
        uint64_t width = n >> (k + 1);               // level stored at nodes[width .. 2*width)
        uint64_t idx = (index >> (k + 1)) ^ 1;
        const uint4 *src = nodes + 2 * (width + idx);
        out[2 * (lane + 1)] = src[0];
        out[2 * (lane + 1) + 1] = src[1];
    }
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
hipError_t iop_query_launch(hipStream_t s, const uint4 *leaf_pair, const uint4 *nodes, uint64_t n,
                            uint64_t index, uint4 *out, const B2Mid &mid)
{
    hipLaunchKernelGGL(k_iop_query, dim3(1), dim3(64), 0, s, leaf_pair, nodes, n, index, out, mid);
    return hipGetLastError();
}

hipError_t merkle_build_launch(hipStream_t s, const uint4 *leafs, uint4 *nodes, uint64_t n,
                               const B2Mid &mid, uint32_t batch)
{
    // n >= 2, power of two (checked by the caller); `batch` trees back to back
    uint64_t m = n;
    bool first = true;
    while (m > 1) {
        uint32_t log_ch = 0;
        while ((1ull << (log_ch + 1)) <= m && log_ch + 1 <= MERKLE_LOG_CH) log_ch++;
        uint64_t chunks = m >> log_ch;
        // A chunk's last levels are narrower than a wave: each is one compression's latency with the
        // rest of the workgroup idle.  When many chunks follow anyway, stop at level width `tail_w`
        // and let the next launch (whose chunks are again full) pick the survivors up.
        static int tail_log = -1;
        if (tail_log < 0) {
            const char *e = getenv("HODOR_MERKLE_TAIL_LOG");
            tail_log = e ? atoi(e) : 6;
        }
        uint32_t levels = log_ch;
        if (chunks >= 512 && log_ch > (uint32_t)tail_log) levels = log_ch - (uint32_t)tail_log;
        if (first)
            hipLaunchKernelGGL(k_merkle_subtree<true>, dim3((unsigned)chunks, batch), dim3(256), 0, s, leafs, nodes,
                               m, log_ch, levels, n, mid);
        else
            hipLaunchKernelGGL(k_merkle_subtree<false>, dim3((unsigned)chunks, batch), dim3(256), 0, s, leafs, nodes,
                               m, log_ch, levels, n, mid);
        m >>= levels;
        first = false;
    }
    return hipGetLastError();
}

hipError_t challenge_launch(hipStream_t s, const uint4 *nodes, uint4 *out, uint4 *root_out, const Fr &r2,
                            uint32_t shave_bits, const FrParams &P)
{
    hipLaunchKernelGGL(k_challenge, dim3(1), dim3(64), 0, s, nodes, out, root_out, r2, shave_bits, P);
    return hipGetLastError();
}

}  // namespace hodor
