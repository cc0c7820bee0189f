// fr9.cuh — K1': carry-free field arithmetic for the NTT inner loops: 9 limbs of 29 bits.
//
// Why: on gfx950 every 32x32 multiply form issues at half the VALU rate and so does every
// carry-in add (bench/microbench.hip), so a 32-bit-limb Montgomery product costs 128 mads PLUS 128
// carry instructions and a field add/sub is a 16-deep VCC chain.  With 29-bit limbs the column sums
// of a product (<= 18 terms of < 2^60) fit a 64-bit accumulator, so a product is 162
// v_mad_u64_u32 and ~45 cheap ops with no carry handling at all, and add/sub are 9 independent
// 32-bit adds.  Values are kept lazily reduced inside a kernel and brought back to the canonical
// 8 x 32-bit Montgomery image (the reference's memory format, R = 2^256) when they leave it.
//
// Representation: value = sum v[i] * 2^(29 i).  "normalized" = every limb < 2^29.
// Montgomery radix here is R' = 2^261.  Data stays in the reference's R-form (x * 2^256); every
// multiplication in the transforms is data x table-constant, and the constants are stored in
// R'-form (w * 2^261), so  (x 2^256)(w 2^261) / 2^261 = (x w) 2^256  — no conversion of the data.
//
// Bounds (p < 2^255):
//   fr9_mul(a, b): a may be lazy (limbs < 2^31.5, value < 2^261), b normalized with value < 2p;
//                  result normalized, value < a*b/2^261 + p  (< 1.5 p for a < 2^260, b < p).
//   fr9_add: limb-wise, no reduction.   fr9_sub(a, b) = a + C - b with C = 4p spread so that every
//                  low limb of C is >= 2^29 (b must be normalized, value < 4p).
#pragma once
#include "fr.cuh"

namespace hodor {

#define HODOR_M29 0x1fffffffu

struct Fr9 {
    uint32_t v[9];
};

struct Fr9Params {      // kernel argument -> SGPRs
    uint32_t p[9];      // modulus, normalized 29-bit limbs
    uint32_t pinv;      // -p^-1 mod 2^29
    uint32_t c4p[9];    // 4p with limbs 0..7 in [2^29, 2^30): subtraction offset (subtrahend < 2.1p)
    uint32_t c5p[9];    // 5p spread the same way: subtraction offset of k_ntt_pass (subtrahend < 4p)
    uint32_t c11p[9];   // 11p spread the same way: offset for subtrahends < 10p (W9 products, fr9w3.cuh)
    uint32_t mu;        // floor(2^(red_bit + 16) / p): quotient estimate for the partial reduction
    uint32_t red_shift; // red_bit - 232 with red_bit = NUM_BITS - 5: the estimate reads x >> red_bit from limb 8
};

// ---- format conversion: 8 x 32-bit words <-> 9 x 29-bit limbs (same integer) ----
__device__ __forceinline__ Fr9 fr9_unpack(const Fr &a)
{
    Fr9 r;
    r.v[0] = a.v[0] & HODOR_M29;
    r.v[1] = ((a.v[0] >> 29) | (a.v[1] << 3)) & HODOR_M29;
    r.v[2] = ((a.v[1] >> 26) | (a.v[2] << 6)) & HODOR_M29;
    r.v[3] = ((a.v[2] >> 23) | (a.v[3] << 9)) & HODOR_M29;
    r.v[4] = ((a.v[3] >> 20) | (a.v[4] << 12)) & HODOR_M29;
    r.v[5] = ((a.v[4] >> 17) | (a.v[5] << 15)) & HODOR_M29;
    r.v[6] = ((a.v[5] >> 14) | (a.v[6] << 18)) & HODOR_M29;
    r.v[7] = ((a.v[6] >> 11) | (a.v[7] << 21)) & HODOR_M29;
    r.v[8] = a.v[7] >> 8;
    return r;
}

// requires normalized limbs and value < 2^256
__device__ __forceinline__ Fr fr9_pack(const Fr9 &a)
{
    Fr r;
    r.v[0] = a.v[0] | (a.v[1] << 29);
    r.v[1] = (a.v[1] >> 3) | (a.v[2] << 26);
    r.v[2] = (a.v[2] >> 6) | (a.v[3] << 23);
    r.v[3] = (a.v[3] >> 9) | (a.v[4] << 20);
    r.v[4] = (a.v[4] >> 12) | (a.v[5] << 17);
    r.v[5] = (a.v[5] >> 15) | (a.v[6] << 14);
    r.v[6] = (a.v[6] >> 18) | (a.v[7] << 11);
    r.v[7] = (a.v[7] >> 21) | (a.v[8] << 8);
    return r;
}

// carry propagation of lazy (unsigned, < 2^32) limbs
__device__ __forceinline__ void fr9_normalize(Fr9 &a)
{
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t c = a.v[i] >> 29;
        a.v[i] &= HODOR_M29;
        a.v[i + 1] += c;
    }
}

// borrow/carry propagation when limbs are signed (after a limb-wise subtraction)
__device__ __forceinline__ void fr9_normalize_signed(Fr9 &a)
{
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int32_t c = (int32_t)a.v[i] >> 29;
        a.v[i] &= HODOR_M29;
        a.v[i + 1] += (uint32_t)c;
    }
}

__device__ __forceinline__ Fr9 fr9_add(const Fr9 &a, const Fr9 &b)
{
    Fr9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}

// a - b + 4p, limb-wise non-negative when b is normalized and < 4p
__device__ __forceinline__ Fr9 fr9_sub(const Fr9 &a, const Fr9 &b, const Fr9Params &P)
{
    Fr9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + (P.c4p[i] - b.v[i]);
    return r;
}

// Montgomery product a * b / 2^261 mod p (not fully reduced, see header): one v_mad_u64_u32 per limb
// product (162) and no carry instructions.
// The empty asm pins the accumulator after every product.  Left alone, the compiler sums each column
// in its own chain starting from 0 and joins the carry of the previous column with a separate 64-bit
// add (v_lshl_add_u64 — half rate, like the mads): 19 extra issue slots per product for instruction
// level parallelism this kernel does not need (the other waves of the SIMD hide the mad latency).
// Pinned: 162 mads in one dependent chain, +6 % products/s in bench/microbench.hip.
#define FR9_MAD(acc, x, y) do { acc += (uint64_t)(x) * (y); asm("" : "+v"(acc)); } while (0)
__device__ __forceinline__ Fr9 fr9_mul(const Fr9 &a, const Fr9 &b, const Fr9Params &P)
{
    uint32_t m[9];
    Fr9 t;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int j = 0; j <= k; j++) FR9_MAD(acc, a.v[j], b.v[k - j]);
#pragma unroll
        for (int j = 0; j < k; j++) FR9_MAD(acc, m[j], P.p[k - j]);
        m[k] = ((uint32_t)acc * P.pinv) & HODOR_M29;
        FR9_MAD(acc, m[k], P.p[0]);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int j = k - 8; j < 9; j++) FR9_MAD(acc, a.v[j], b.v[k - j]);
#pragma unroll
        for (int j = k - 8; j < 9; j++) FR9_MAD(acc, m[j], P.p[k - j]);
        t.v[k - 9] = (uint32_t)acc & HODOR_M29;
        acc >>= 29;
    }
    t.v[8] = (uint32_t)acc;
    return t;
}

// Bring a lazy value (limbs < 2^32, value < 2^261) into [0, 2p) by subtracting q*p with
//   q = floor(floor(x / 2^b) * mu / 2^16),  b = NUM_BITS - 5,  mu = floor(2^(b+16) / p).
// q never exceeds floor(x/p), and q >= x/p - x/2^(b+16) - 2^b/p > x/p - 1/32 - 1/16, hence
// q >= floor(x/p) - 1 and the remainder is < 2p (one conditional subtraction away from canonical).
// Result normalized.  (NUM_BITS >= 240 is checked at context creation, so bit b lies in limb 8.)
template <bool NORMALIZED = false>   // NORMALIZED: limbs 0..7 are already < 2^29 (skip the first carry pass)
__device__ __forceinline__ void fr9_reduce_partial(Fr9 &a, const Fr9Params &P)
{
    if (!NORMALIZED) fr9_normalize(a);
    uint32_t q = ((a.v[8] >> P.red_shift) * P.mu) >> 16;   // limb 8 holds bits 232 and up
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t s = (uint64_t)q * P.p[i] + carry;
        a.v[i] -= (uint32_t)s & HODOR_M29;
        carry = s >> 29;
    }
    a.v[8] -= (uint32_t)((uint64_t)q * P.p[8] + carry);   // q*p <= x, so this limb fits
    fr9_normalize_signed(a);
}

// conditional subtraction: a in [0, 2^30 * ...) normalized; returns a - p if a >= p else a
__device__ __forceinline__ void fr9_cond_sub_p(Fr9 &a, const Fr9Params &P)
{
    Fr9 d;
#pragma unroll
    for (int i = 0; i < 9; i++) d.v[i] = a.v[i] - P.p[i];
    fr9_normalize_signed(d);
    bool neg = (int32_t)d.v[8] < 0;
#pragma unroll
    for (int i = 0; i < 9; i++) a.v[i] = neg ? a.v[i] : d.v[i];
}

// lazy value -> canonical [0, p) in the 8 x 32-bit memory format
template <bool NORMALIZED = false>
__device__ __forceinline__ Fr fr9_to_canonical(Fr9 a, const Fr9Params &P)
{
    fr9_reduce_partial<NORMALIZED>(a, P);      // < 2p
    fr9_cond_sub_p(a, P);
    return fr9_pack(a);
}

// lazy value -> some representative < 2^256 in the 8 x 32-bit memory format (between passes)
template <bool NORMALIZED = false>
__device__ __forceinline__ Fr fr9_to_packed(Fr9 a, const Fr9Params &P)
{
    fr9_reduce_partial<NORMALIZED>(a, P);
    return fr9_pack(a);
}

// ---- table entries: 9 limbs stored in 12 words (48 B = 3 x dwordx4) ----
__device__ __forceinline__ Fr9 fr9_load48(const void *ptr)
{
    const uint4 *q = reinterpret_cast<const uint4 *>(ptr);
    uint4 a = q[0], b = q[1], c = q[2];
    Fr9 r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    r.v[8] = c.x;
    return r;
}

__device__ __forceinline__ void fr9_store48(void *ptr, const Fr9 &a)
{
    uint4 *q = reinterpret_cast<uint4 *>(ptr);
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
    q[2] = make_uint4(a.v[8], 0, 0, 0);
}

}  // namespace hodor
