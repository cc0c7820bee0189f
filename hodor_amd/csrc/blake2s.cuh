// blake2s.cuh — device-side keyed BLAKE2s for the Merkle kernels (merkle.hip) and the fused FRI tail
// (fri.hip).  Parameters as in /root/reference/src/iop/blake2s_trivial_iop.rs:8-16; the keyed first
// block's chaining value (B2Mid) is precomputed on the host, so a leaf or node hash is ONE compression.
// 32-bit ARX integer work: one hash per lane, 16 state + 16 message words in VGPRs, sigma schedule
// resolved at compile time.
#pragma once
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
__device__ __forceinline__ void b2s_final(const uint32_t (&h0)[8], const uint32_t m[16], uint32_t t,
                                          uint32_t out[8])
{
    uint32_t v0 = h0[0], v1 = h0[1], v2 = h0[2], v3 = h0[3];
    uint32_t v4 = h0[4], v5 = h0[5], v6 = h0[6], v7 = h0[7];
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
    out[0] = h0[0] ^ v0 ^ v8;  out[1] = h0[1] ^ v1 ^ v9;
    out[2] = h0[2] ^ v2 ^ v10; out[3] = h0[3] ^ v3 ^ v11;
    out[4] = h0[4] ^ v4 ^ v12; out[5] = h0[5] ^ v5 ^ v13;
    out[6] = h0[6] ^ v6 ^ v14; out[7] = h0[7] ^ v7 ^ v15;
}

// hash of one 32-byte leaf (message words 8..15 are zero and fold away at compile time)
__device__ __forceinline__ void b2s_leaf(const B2Mid &mid, const uint4 &lo, const uint4 &hi,
                                         uint32_t out[8])
{
    uint32_t m[16] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w, 0, 0, 0, 0, 0, 0, 0, 0};
    b2s_final(mid.h, m, 96, out);
}

// hash of one COSET2 leaf: the 64 bytes of two field elements — one compression like a node, but on top of the
// "Shaftoe2" midstate (mid.hp): the two child digests of an interior node never hash to that node as a "leaf"
__device__ __forceinline__ void b2s_pair(const B2Mid &mid, const uint4 &a0, const uint4 &a1, const uint4 &b0,
                                         const uint4 &b1, uint32_t out[8])
{
    uint32_t m[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    b2s_final(mid.hp, m, 128, out);
}

__device__ __forceinline__ void b2s_node(const B2Mid &mid, const uint32_t l[8], const uint32_t r[8],
                                         uint32_t out[8])
{
    uint32_t m[16];
#pragma unroll
    for (int i = 0; i < 8; i++) { m[i] = l[i]; m[8 + i] = r[i]; }
    b2s_final(mid.h, m, 128, out);
}

// ---------------------------------------------------------------------------------------------
// Quad-lane compression: FOUR lanes share one hash.  Lane j of a quad holds column j of the 4 x 4
// state (a_j, b_j, c_j, d_j), so a round is two G's per lane instead of eight and the dependent
// instruction chain of a compression shrinks ~3.3x.  The diagonal step reads rows b, c, d from the
// neighbouring lanes with DPP quad permutes.  It costs more instructions in total (message words are
// fetched per lane from LDS, state rotations), so it is for the narrow levels at the top of a tree and
// for small trees, where lanes are idle and the tree's depth — one compression latency per level — is
// what the commit waits for.  All four lanes of a quad must be active.
// ---------------------------------------------------------------------------------------------
struct B2Quad {
    uint32_t off[40];                     // byte offset, in use order, of the message words this lane feeds to G
    uint32_t a0, b0, c0, d0_leaf, d0_node;   // this lane's column of the initial state (t = 96 / 128, final block)
    uint32_t a0p, b0p;                       // rows a, b of the COSET2 leaf midstate (mid.hp)
};

__device__ __forceinline__ uint32_t b2q_pick(uint32_t j, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3)
{
    return j == 0 ? x0 : (j == 1 ? x1 : (j == 2 ? x2 : x3));
}

__device__ __forceinline__ void b2q_init(B2Quad &q, const B2Mid &mid, uint32_t j /* lane & 3 */)
{
    constexpr uint8_t S[10][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
#pragma unroll
    for (int r = 0; r < 10; r++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            // i = 0,1: column step (words 2j, 2j+1 of the schedule); i = 2,3: diagonal step (8+2j, 9+2j)
            const int base = (i < 2 ? 0 : 8) + (i & 1);
            q.off[4 * r + i] = b2q_pick(j, 4u * S[r][base], 4u * S[r][base + 2], 4u * S[r][base + 4], 4u * S[r][base + 6]);
        }
    }
    q.a0 = b2q_pick(j, mid.h[0], mid.h[1], mid.h[2], mid.h[3]);
    q.b0 = b2q_pick(j, mid.h[4], mid.h[5], mid.h[6], mid.h[7]);
    q.a0p = b2q_pick(j, mid.hp[0], mid.hp[1], mid.hp[2], mid.hp[3]);
    q.b0p = b2q_pick(j, mid.hp[4], mid.hp[5], mid.hp[6], mid.hp[7]);
    q.c0 = b2q_pick(j, 0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au);
    q.d0_leaf = b2q_pick(j, 0x510E527Fu ^ 96u, 0x9B05688Cu, ~0x1F83D9ABu, 0x5BE0CD19u);
    q.d0_node = b2q_pick(j, 0x510E527Fu ^ 128u, 0x9B05688Cu, ~0x1F83D9ABu, 0x5BE0CD19u);
}

#define B2Q_DPP(x, ctrl) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), (ctrl), 0xf, 0xf, true))

// `msg`: the 64-byte message block in LDS (the same address in the four lanes).  Returns this lane's
// two digest words: h_lo = word j, h_hi = word 4 + j.  kind: B2Q_LEAF (32-byte leaf, t = 96), B2Q_NODE (t = 128),
// B2Q_PAIR (a COSET2 leaf: t = 128 on the "Shaftoe2" midstate).
enum { B2Q_LEAF = 0, B2Q_NODE = 1, B2Q_PAIR = 2 };
__device__ __forceinline__ void b2q_compress(const B2Quad &q, const uint32_t *msg, int kind, uint32_t &h_lo,
                                             uint32_t &h_hi)
{
    const bool node = kind != B2Q_LEAF;
    const uint32_t ia = kind == B2Q_PAIR ? q.a0p : q.a0, ib = kind == B2Q_PAIR ? q.b0p : q.b0;
    uint32_t m[40];
#pragma unroll
    for (int i = 0; i < 40; i++)
        m[i] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(msg) + q.off[i]);
    uint32_t a = ia, b = ib, c = q.c0, d = node ? q.d0_node : q.d0_leaf;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        if (r) {   // back from the diagonal layout: column j's b, c, d sit in lanes j+3, j+2, j+1
            b = B2Q_DPP(b, 0x93); c = B2Q_DPP(c, 0x4E); d = B2Q_DPP(d, 0x39);
        }
        B2S_G(a, b, c, d, m[4 * r], m[4 * r + 1]);
        b = B2Q_DPP(b, 0x39); c = B2Q_DPP(c, 0x4E); d = B2Q_DPP(d, 0x93);   // lane j: b_(j+1), c_(j+2), d_(j+3)
        B2S_G(a, b, c, d, m[4 * r + 2], m[4 * r + 3]);
    }
    b = B2Q_DPP(b, 0x93); c = B2Q_DPP(c, 0x4E); d = B2Q_DPP(d, 0x39);
    h_lo = ia ^ a ^ c;
    h_hi = ib ^ b ^ d;
}

// interpret_hash (src/iop/blake2s_trivial_iop.rs:48-60): big-endian read of the digest words `d`,
// clear the top 256 - CAPACITY bits, convert to Montgomery form (multiply by R^2).
__device__ __forceinline__ Fr b2s_digest_to_challenge(const uint32_t d[8], const Fr &r2, uint32_t shave_bits,
                                                      const FrParams &P)
{
    Fr x;
#pragma unroll
    for (int i = 0; i < 8; i++) x.v[i] = __builtin_bswap32(d[7 - i]);   // byte-reverse 32 bytes
    uint32_t s = shave_bits & 63;   // the reference shifts a 64-bit mask by SHAVE_BITS % 64
    uint64_t mask = ~0ull >> s;
    x.v[7] &= (uint32_t)(mask >> 32);
    x.v[6] &= (uint32_t)mask;
    return fr_mul(x, r2, P);
}

}  // namespace hodor
