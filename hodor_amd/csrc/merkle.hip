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
// 32-bit ARX integer work: no MFMA.  The compression functions (one hash per lane, and the quad-lane
// variant for narrow levels) live in blake2s.cuh; this file holds the two tree schedules.
#include <cstdlib>

#include "blake2s.cuh"
#include "knobs.hpp"
#include "ntt.cuh"

namespace hodor {

// ---------------------------------------------------------------------------------------------
// k_merkle_subtree: one workgroup builds the complete subtree over a chunk of CH = 2^log_ch inputs
// (leaves when LEAF, digests of tree level `m` otherwise): every level of the chunk is written to its
// place in the heap array and handed to the next level through LDS, so each leaf / digest is read
// from HBM once and a 2^25-leaf tree takes 5 launches instead of 25.
//
//   level k of the chunk (k = 1 .. log_ch) has CH >> k nodes at nodes[(m >> k) + chunk*(CH >> k) ..)
//
// Two schedules:
//   throughput (LAT = false, big levels): 1024 inputs per workgroup; phase 0 pairs two adjacent inputs
//     per lane (64 contiguous bytes; a wave reads 4 KiB contiguous) and hashes leaf, leaf, node in one
//     go; later levels re-map the surviving nodes densely onto the lanes.  The launcher stops a chunk
//     before its levels get narrower than a wave.
//   latency (LAT = true, levels that cannot fill the chip anyway): <= 256 inputs per workgroup, one
//     input per lane first (a leaf hash, or a plain copy into LDS), then one tree level per step with
//     the quad-lane compression (blake2s.cuh) as soon as a level has fewer nodes than the workgroup
//     has quads — the commit waits for depth x compression latency here, not for throughput.
// ---------------------------------------------------------------------------------------------
#ifndef HODOR_MERKLE_LOG_CH
#define HODOR_MERKLE_LOG_CH 10
#endif
constexpr uint32_t MERKLE_LOG_CH = HODOR_MERKLE_LOG_CH;   // throughput: 1024 inputs per workgroup, 16 KiB + 8 KiB of LDS (6 workgroups per CU)
constexpr uint32_t MERKLE_LAT_LOG_CH = 8;    // latency: 256 inputs per workgroup

// FOLD (leaf launch only, either schedule): the leaves do not exist yet — leaf i is the FRI fold of the
// previous round's values (fri_fold_one), computed here, stored to `fold.dst` and hashed from registers,
// which saves the separate fold launch and its round trip through memory.
// COMB (leaf launch only): the COSET2 combiner — leaf k of the tree is the 64-byte pair value[k] || value[k + n/2]
// (the coset FRI opens together, src/fri/query_producer.rs:27-34; CosetCombiner seam src/iop/mod.rs:22-34), hashed
// with ONE compression; the tree has m = n/2 leaves and its heap array n/2 entries.  With FOLD the two values of a
// leaf are fold outputs k and k + n/2 of the round.
template <bool LEAF, bool LAT, bool FOLD = false, bool COMB = false>
__global__ void __launch_bounds__(256)
k_merkle_subtree(const uint4 *leafs, uint4 *nodes, uint64_t m, uint32_t log_ch, uint32_t levels, uint64_t n,
                 B2Mid mid, FoldArgs fold = FoldArgs(), Fr9Params Q = Fr9Params() BXPARAM_DEF)
{
#ifdef HODOR_BOUNDS
    // the buffers as the launcher declared them, and where this tree's arrays start in them (bounds.cuh)
    const uint4 *const leafs0 = leafs;
    uint4 *const nodes0 = nodes;
    const uint64_t leaf_off = 2 * (uint64_t)blockIdx.y * n, node_off = 2 * (uint64_t)blockIdx.y * (COMB ? n >> 1 : n);
    const uint4 *const in_base = LEAF ? leafs0 : (const uint4 *)nodes0;
    const uint64_t in_off = LEAF ? leaf_off + 2 * ((uint64_t)blockIdx.x << log_ch) : node_off + 2 * (m + ((uint64_t)blockIdx.x << log_ch));
#endif
    // blockIdx.y selects one of several independent trees over n values each (batched commit)
    leafs += 2 * (uint64_t)blockIdx.y * n;
    nodes += 2 * (uint64_t)blockIdx.y * (COMB ? n >> 1 : n);
    const uint64_t half = n >> 1;   // COMB: distance between the two values of a leaf
    // throughput: level 1 (1024 digests) | level 2 (512); latency: the inputs (256 digests) | level 1 (128)
    constexpr uint32_t CAP_A = LAT ? (1u << MERKLE_LAT_LOG_CH) : (1u << (MERKLE_LOG_CH - 1));
    __shared__ uint4 buf_a[2 * CAP_A];
    __shared__ uint4 buf_b[CAP_A];
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
    const uint32_t ch = 1u << log_ch;
    const uint64_t chunk = blockIdx.x;
    const uint4 *in = LEAF ? leafs + 2 * (chunk << log_ch) : nodes + 2 * (m + (chunk << log_ch));
    uint4 *src = buf_a, *dst = buf_b;
    uint32_t k0;

    if (LAT) {
        // one input per lane into LDS: leaf hash (LEAF) or the digest itself
        for (uint32_t p = tid; p < ch; p += nthreads) {
            uint4 a0, a1, b0, b1;
            if (FOLD) {
                const uint64_t i = (chunk << log_ch) + p;
                Fr y = fri_fold_one(fold, i, Q BXPASS);
                fr_store(BATS(1, fold.dst, 2 * i, 2), y);
                a0 = make_uint4(y.v[0], y.v[1], y.v[2], y.v[3]);
                a1 = make_uint4(y.v[4], y.v[5], y.v[6], y.v[7]);
                if (COMB) {
                    y = fri_fold_one(fold, i + half, Q BXPASS);
                    fr_store(BATS(2, fold.dst, 2 * (i + half), 2), y);
                    b0 = make_uint4(y.v[0], y.v[1], y.v[2], y.v[3]);
                    b1 = make_uint4(y.v[4], y.v[5], y.v[6], y.v[7]);
                }
            } else {
                a0 = *BATP(3, in_base, in_off + 2 * p, 2, in + 2 * p);
                a1 = in[2 * p + 1];
                if (COMB) { b0 = *BATP(4, in_base, in_off + 2 * (p + half), 2, in + 2 * (p + half)); b1 = in[2 * (p + half) + 1]; }
            }
            if (LEAF) {
                uint32_t out[8];
                if (COMB) b2s_pair(mid, a0, a1, b0, b1, out);
                else b2s_leaf(mid, a0, a1, out);
                a0 = make_uint4(out[0], out[1], out[2], out[3]);
                a1 = make_uint4(out[4], out[5], out[6], out[7]);
            }
            buf_a[2 * p] = a0; buf_a[2 * p + 1] = a1;
        }
        k0 = 1;
    } else {
        // level 1: pairs of inputs
        uint4 *lvl_out = nodes + 2 * ((m >> 1) + chunk * (ch >> 1));
        for (uint32_t p = tid; p < (ch >> 1); p += nthreads) {
            uint4 a0, a1, b0, b1;
            uint4 c0, c1, d0, d1;   // COMB: the upper halves of the two leaves (values k + n/2, k + 1 + n/2)
            if (FOLD) {
                // the two leaves of this lane are FRI fold outputs that do not exist yet: compute them
                // (two products each), store them as the round's values, hash them from registers
                const uint64_t i = (chunk << log_ch) + 2 * p;
                Fr y0 = fri_fold_one(fold, i, Q BXPASS), y1 = fri_fold_one(fold, i + 1, Q BXPASS);
                fr_store(BATS(5, fold.dst, 2 * i, 2), y0);
                fr_store(BATS(6, fold.dst, 2 * i + 2, 2), y1);
                a0 = make_uint4(y0.v[0], y0.v[1], y0.v[2], y0.v[3]);
                a1 = make_uint4(y0.v[4], y0.v[5], y0.v[6], y0.v[7]);
                b0 = make_uint4(y1.v[0], y1.v[1], y1.v[2], y1.v[3]);
                b1 = make_uint4(y1.v[4], y1.v[5], y1.v[6], y1.v[7]);
                if (COMB) {
                    y0 = fri_fold_one(fold, i + half, Q BXPASS);
                    y1 = fri_fold_one(fold, i + half + 1, Q BXPASS);
                    fr_store(BATS(7, fold.dst, 2 * (i + half), 2), y0);
                    fr_store(BATS(8, fold.dst, 2 * (i + half) + 2, 2), y1);
                    c0 = make_uint4(y0.v[0], y0.v[1], y0.v[2], y0.v[3]);
                    c1 = make_uint4(y0.v[4], y0.v[5], y0.v[6], y0.v[7]);
                    d0 = make_uint4(y1.v[0], y1.v[1], y1.v[2], y1.v[3]);
                    d1 = make_uint4(y1.v[4], y1.v[5], y1.v[6], y1.v[7]);
                }
            } else {
                const uint4 *q = BATP(9, in_base, in_off + 4 * p, 4, in + 4 * p);
                a0 = q[0]; a1 = q[1]; b0 = q[2]; b1 = q[3];
                if (COMB) { q = BATP(10, in_base, in_off + 4 * p + 2 * half, 4, q + 2 * half); c0 = q[0]; c1 = q[1]; d0 = q[2]; d1 = q[3]; }
            }
            uint32_t l[8], r[8], out[8];
            if (LEAF && COMB) {
                b2s_pair(mid, a0, a1, c0, c1, l);
                b2s_pair(mid, b0, b1, d0, d1, r);
            } else if (LEAF) {
                b2s_leaf(mid, a0, a1, l);
                b2s_leaf(mid, b0, b1, r);
            } else {
                l[0] = a0.x; l[1] = a0.y; l[2] = a0.z; l[3] = a0.w; l[4] = a1.x; l[5] = a1.y; l[6] = a1.z; l[7] = a1.w;
                r[0] = b0.x; r[1] = b0.y; r[2] = b0.z; r[3] = b0.w; r[4] = b1.x; r[5] = b1.y; r[6] = b1.z; r[7] = b1.w;
            }
            b2s_node(mid, l, r, out);
            uint4 o0 = make_uint4(out[0], out[1], out[2], out[3]), o1 = make_uint4(out[4], out[5], out[6], out[7]);
            *BATSP(11, nodes0, node_off + 2 * ((m >> 1) + chunk * (ch >> 1)) + 2 * p, 2, lvl_out + 2 * p) = o0; lvl_out[2 * p + 1] = o1;
            buf_a[2 * p] = o0; buf_a[2 * p + 1] = o1;
        }
        k0 = 2;
    }
    __syncthreads();

    B2Quad bq;
    if (LAT) b2q_init(bq, mid, tid & 3);
    // remaining levels: ping-pong between the two LDS buffers
    for (uint32_t k = k0; k <= levels; k++) {
        const uint32_t w = ch >> k;
        uint4 *lvl_out = nodes + 2 * ((m >> k) + chunk * w);
        if (LAT && 4 * w <= nthreads) {
            const uint32_t quad = tid >> 2, j = tid & 3;
            if (quad < w) {
                uint32_t lo, hi;
                b2q_compress(bq, reinterpret_cast<const uint32_t *>(src + 4 * quad), B2Q_NODE, lo, hi);
                uint32_t *o = reinterpret_cast<uint32_t *>(dst + 2 * quad);
                o[j] = lo; o[4 + j] = hi;
                uint32_t *g = reinterpret_cast<uint32_t *>(BATSP(12, nodes0, node_off + 2 * ((m >> k) + chunk * w) + 2 * quad, 2, lvl_out + 2 * quad));
                g[j] = lo; g[4 + j] = hi;
            }
        } else {
            for (uint32_t g = tid; g < w; g += nthreads) {
                uint4 a0 = src[4 * g], a1 = src[4 * g + 1], b0 = src[4 * g + 2], b1 = src[4 * g + 3];
                uint32_t l[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                uint32_t r[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                uint32_t out[8];
                b2s_node(mid, l, r, out);
                uint4 o0 = make_uint4(out[0], out[1], out[2], out[3]), o1 = make_uint4(out[4], out[5], out[6], out[7]);
                *BATSP(13, nodes0, node_off + 2 * ((m >> k) + chunk * w) + 2 * g, 2, lvl_out + 2 * g) = o0; lvl_out[2 * g + 1] = o1;
                dst[2 * g] = o0; dst[2 * g + 1] = o1;
            }
        }
        __syncthreads();
        uint4 *t = src; src = dst; dst = t;
    }
    // the launch that produces the root also clears nodes[0], which the heap layout leaves unused
    if ((m >> levels) == 1 && tid == 0) {
        *BATSP(14, nodes0, node_off, 2, nodes) = make_uint4(0, 0, 0, 0);
        nodes[1] = make_uint4(0, 0, 0, 0);
    }
}

// K8: root digest -> field challenge (interpret_hash): big-endian read, clear the top
// 256 - CAPACITY bits, convert to Montgomery form (multiply by R^2).
// `root_out` (optional) receives a copy of the root digest: the FRI round loop collects its roots there.
__global__ void k_challenge(const uint4 *nodes, uint4 *out, uint4 *root_out, Fr r2, uint32_t shave_bits,
                            FrParams P BXPARAM)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (root_out) {
        *BATS(1, root_out, 0, 2) = nodes[2];
        root_out[1] = nodes[3];
    }
    const uint32_t *d = reinterpret_cast<const uint32_t *>(BAT(2, nodes, 2, 2));   // nodes[1]
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = d[i];
    fr_store(BATS(3, out, 0, 2), b2s_digest_to_challenge(w, r2, shave_bits, P));
}

// Query phase on device-resident oracles: IOP::query + IopTree::get_path
// (src/iop/blake2s_trivial_iop.rs:251-279, 324-338).  out[0] = the queried leaf value,
// out[1] = hash of its sibling leaf, out[1 + k] = sibling node at tree level log2(n) - 1 - k.
// `leaf_pair` points at the even leaf of the pair {index & ~1, index | 1}.
__global__ void k_iop_query(const uint4 *leaf_pair, const uint4 *nodes, uint64_t n, uint64_t index,
                            uint4 *out, B2Mid mid BXPARAM)
{
    const uint32_t lane = threadIdx.x;
    uint32_t levels = 0;
    for (uint64_t w = n >> 1; w >= 2; w >>= 1) levels++;   // log2(n) - 1 node levels on the path
    if (lane == 0) {
        const uint4 *me = BAT(1, leaf_pair, 2 * (index & 1), 2), *sib = BAT(2, leaf_pair, 2 * ((index & 1) ^ 1), 2);
        *BATS(3, out, 0, 4) = me[0];
        out[1] = me[1];
        uint32_t h[8];
        b2s_leaf(mid, sib[0], sib[1], h);
        out[2] = make_uint4(h[0], h[1], h[2], h[3]);
        out[3] = make_uint4(h[4], h[5], h[6], h[7]);
    } else if (lane <= levels) {
        uint32_t k = lane - 1;                       // k-th node level from the bottom
        uint64_t width = n >> (k + 1);               // level stored at nodes[width .. 2*width)
        uint64_t idx = (index >> (k + 1)) ^ 1;
        const uint4 *src = BAT(4, nodes, 2 * (width + idx), 2);
        *BATS(5, out, 2 * (lane + 1), 2) = src[0];
        out[2 * (lane + 1) + 1] = src[1];
    }
}

// The same for a COSET2 tree (n values, n/2 leaves): out[0] = value[k], out[1] = value[k + n/2] for the leaf
// k = index mod n/2, out[2] = hash of the sibling leaf, out[2 + j] = sibling node at tree level log2(n/2) - 1 - j.
__global__ void k_iop_query_coset2(const uint4 *values, const uint4 *nodes, uint64_t n, uint64_t index, uint4 *out,
                                   B2Mid mid BXPARAM)
{
    const uint32_t lane = threadIdx.x;
    const uint64_t leaves = n >> 1, k = index & (leaves - 1);
    uint32_t levels = 0;
    for (uint64_t w = leaves >> 1; w >= 2; w >>= 1) levels++;   // log2(leaves) - 1 node levels on the path
    if (lane == 0) {
        const uint4 *lo = BAT(1, values, 2 * k, 2), *hi = BAT(2, values, 2 * (k + leaves), 2);
        *BATS(3, out, 0, 6) = lo[0]; out[1] = lo[1];
        out[2] = hi[0]; out[3] = hi[1];
        const uint4 *slo = BAT(4, values, 2 * (k ^ 1), 2), *shi = BAT(5, values, 2 * ((k ^ 1) + leaves), 2);
        uint32_t h[8];
        b2s_pair(mid, slo[0], slo[1], shi[0], shi[1], h);
        out[4] = make_uint4(h[0], h[1], h[2], h[3]);
        out[5] = make_uint4(h[4], h[5], h[6], h[7]);
    } else if (lane <= levels) {
        uint32_t j = lane - 1;
        uint64_t width = leaves >> (j + 1);
        uint64_t idx = (k >> (j + 1)) ^ 1;
        const uint4 *src = BAT(6, nodes, 2 * (width + idx), 2);
        *BATS(7, out, 2 * (lane + 2), 2) = src[0];
        out[2 * (lane + 2) + 1] = src[1];
    }
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
hipError_t iop_query_coset2_launch(hipStream_t s, const uint4 *values, const uint4 *nodes, uint64_t n,
                                   uint64_t index, uint4 *out, const B2Mid &mid)
{
    BX_BEGIN(bx, KID_IOP_QUERY_COSET2);
    BX_ADD(bx, values, n * 32);
    BX_ADD(bx, nodes, (n / 2) * 32);
    BX_ADD(bx, out, (uint64_t)(64 - __builtin_clzll(n)) * 32);   // two values + log2(n) - 1 digests = log2(n) + 1 entries
    hipLaunchKernelGGL(k_iop_query_coset2, dim3(1), dim3(64), 0, s, values, nodes, n, index, out, mid BXARG(bx));
    return hipGetLastError();
}

hipError_t iop_query_launch(hipStream_t s, const uint4 *leaf_pair, const uint4 *nodes, uint64_t n,
                            uint64_t index, uint4 *out, const B2Mid &mid)
{
    BX_BEGIN(bx, KID_IOP_QUERY);
    BX_ADD(bx, leaf_pair, 64);
    BX_ADD(bx, nodes, n * 32);
    BX_ADD(bx, out, (uint64_t)(64 - __builtin_clzll(n)) * 32);   // the value + log2(n) digests = log2(n) + 1 entries
    hipLaunchKernelGGL(k_iop_query, dim3(1), dim3(64), 0, s, leaf_pair, nodes, n, index, out, mid BXARG(bx));
    return hipGetLastError();
}

// `fold` (optional, batch == 1): the leaves are fold->dst, still to be computed from fold->src; when the
// tree is small enough for the latency schedule the first launch folds and hashes in one go, otherwise
// the caller must have run the fold already (merkle_fuses_fold tells which).
#define g_tail_log (knobs().merkle_tail_log)
#define g_lat_log (knobs().merkle_lat_log)
static void merkle_knobs() {}
bool merkle_fuses_fold(uint64_t n)
{
    // fri_fuse_fold: 0 never, 1 every round (both schedules), 2 only the latency-schedule rounds
    const int f = knobs().fri_fuse_fold;
    return f == 1 || (f == 2 && n <= (1ull << g_lat_log));
}

hipError_t merkle_build_launch(hipStream_t s, const uint4 *leafs, uint4 *nodes, uint64_t n,
                               const B2Mid &mid, uint32_t batch, const FoldArgs *fold, const Fr9Params *Q, bool comb)
{
    // n >= 2 (comb: n >= 4), power of two (checked by the caller); `batch` trees back to back
    merkle_knobs();
    const int tail_log = g_tail_log, lat_log = g_lat_log;
    uint64_t m = comb ? n >> 1 : n;   // leaves of the tree
    bool first = true;
    while (m > 1) {
        uint32_t log_m = 0;
        while ((1ull << (log_m + 1)) <= m) log_m++;
        // levels of at most 2^lat_log inputs (per tree) cannot fill the chip: latency schedule
        const bool lat = log_m <= (uint32_t)lat_log;
        uint32_t log_ch = log_m < (lat ? MERKLE_LAT_LOG_CH : MERKLE_LOG_CH) ? log_m : (lat ? MERKLE_LAT_LOG_CH : MERKLE_LOG_CH);
        uint64_t chunks = m >> log_ch;
        // A chunk's last levels are narrower than a wave: each is one compression's latency with the
        // rest of the workgroup idle.  When many chunks follow anyway, stop at level width 2^tail_log
        // and let the next launch (whose chunks are again full) pick the survivors up.
        uint32_t levels = log_ch;
        if (!lat && chunks >= 512 && log_ch > (uint32_t)tail_log) levels = log_ch - (uint32_t)tail_log;
        unsigned threads = 256;
        if (lat) {   // one lane per input, at least one wave
            threads = 1u << log_ch;
            if (threads < 64) threads = 64;
        }
        dim3 grid((unsigned)chunks, batch);
        BX_BEGIN(bx, KID_MERKLE_SUBTREE);
        BX_ADD(bx, leafs, (uint64_t)batch * n * 32);
        BX_ADD(bx, nodes, (uint64_t)batch * (comb ? n >> 1 : n) * 32);
#ifdef HODOR_BOUNDS
        if (fold) {
            bx.add(fold->src, 2 * fold->half * 32).add(fold->dst, fold->half * 32).add(fold->lo, fold->lo_bytes).add(fold->hi_beta, fold->hi_bytes);
        }
#endif
        if (first && comb) {
            const FoldArgs fa = fold ? *fold : FoldArgs();
            const Fr9Params qa = Q ? *Q : Fr9Params();
            if (lat && fold)  hipLaunchKernelGGL((k_merkle_subtree<true, true, true, true>), grid, dim3(threads), 0, s, leafs, nodes, m, log_ch, levels, n, mid, fa, qa BXARG(bx));
            else if (lat)     hipLaunchKernelGGL((k_merkle_subtree<true, true, false, true>), grid, dim3(threads), 0, s, leafs, nodes, m, log_ch, levels, n, mid, fa, qa BXARG(bx));
            else if (fold)    hipLaunchKernelGGL((k_merkle_subtree<true, false, true, true>), grid, dim3(threads), 0, s, leafs, nodes, m, log_ch, levels, n, mid, fa, qa BXARG(bx));
            else              hipLaunchKernelGGL((k_merkle_subtree<true, false, false, true>), grid, dim3(threads), 0, s, leafs, nodes, m, log_ch, levels, n, mid, fa, qa BXARG(bx));
        } else if (first && lat && fold)
            hipLaunchKernelGGL((k_merkle_subtree<true, true, true>), grid, dim3(threads), 0, s, leafs, nodes, m, log_ch, levels, n, mid, *fold, *Q BXARG(bx));
        else if (first && lat)
            hipLaunchKernelGGL((k_merkle_subtree<true, true>), grid, dim3(threads), 0, s, leafs, nodes, m, log_ch, levels, n, mid, FoldArgs(), Fr9Params() BXARG(bx));
        else if (first && fold)
            hipLaunchKernelGGL((k_merkle_subtree<true, false, true>), grid, dim3(threads), 0, s, leafs, nodes, m, log_ch, levels, n, mid, *fold, *Q BXARG(bx));
        else if (first)
            hipLaunchKernelGGL((k_merkle_subtree<true, false>), grid, dim3(threads), 0, s, leafs, nodes, m, log_ch, levels, n, mid, FoldArgs(), Fr9Params() BXARG(bx));
        // (the launches above the leaves use `n` only as the distance between the batch's node arrays: n/2 entries for a
        // COSET2 tree)
        else if (lat)
            hipLaunchKernelGGL((k_merkle_subtree<false, true>), grid, dim3(threads), 0, s, leafs, nodes, m, log_ch, levels, comb ? n >> 1 : n, mid, FoldArgs(), Fr9Params() BXARG(bx));
        else
            hipLaunchKernelGGL((k_merkle_subtree<false, false>), grid, dim3(threads), 0, s, leafs, nodes, m, log_ch, levels, comb ? n >> 1 : n, mid, FoldArgs(), Fr9Params() BXARG(bx));
        m >>= levels;
        first = false;
    }
    return hipGetLastError();
}

hipError_t challenge_launch(hipStream_t s, const uint4 *nodes, uint4 *out, uint4 *root_out, const Fr &r2,
                            uint32_t shave_bits, const FrParams &P)
{
    BX_BEGIN(bx, KID_CHALLENGE);
    BX_ADD(bx, nodes, 64);       // nodes[0..2): the root is nodes[1]
    BX_ADD(bx, out, 32);
    BX_ADD(bx, root_out, 32);
    hipLaunchKernelGGL(k_challenge, dim3(1), dim3(64), 0, s, nodes, out, root_out, r2, shave_bits, P BXARG(bx));
    return hipGetLastError();
}

}  // namespace hodor
