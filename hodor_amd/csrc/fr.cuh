// fr.cuh — K1: 256-bit Montgomery prime-field arithmetic for gfx950 (device side).
//
// Replaces ff_ce's derived `Fr` ops (mul_assign / add_assign / sub_assign / square / pow) that
// every reference hot loop calls (e.g. /root/reference/src/fft/fft.rs:52-58).  Element format is
// the reference's memory image: Fr(FrRepr([u64;4])), Montgomery form with R = 2^256, little-endian
// limbs, value in [0, p)  (src/bn256.rs:4-7).  On the GPU an element is 8 x 32-bit limbs in VGPRs;
// the modulus lives in SGPRs (kernel argument), so it is a free scalar operand of v_mad_u64_u32.
//
// Integer work only: no MFMA, no floating point.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bounds.cuh"

namespace hodor {

struct FrParams {        // passed by value as a kernel argument -> SGPRs
    uint32_t p[8];       // modulus, little-endian 32-bit limbs
    uint32_t pinv;       // -p^{-1} mod 2^32
    uint32_t one[8];     // R mod p
};

struct Fr {
    uint32_t v[8];
};

// BLAKE2s chaining value after the keyed first block (see merkle.hip)
struct B2Mid {
    uint32_t h[8];    // chaining value after the key block, personal "Shaftoe": leaves and nodes of the reference's format
    uint32_t hp[8];   // the same with personal "Shaftoe2": COSET2 LEAVES only (round 5: a 64-byte leaf must not hash like a node)
};

__device__ __forceinline__ Fr fr_zero()
{
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
}

__device__ __forceinline__ Fr fr_one(const FrParams &P)
{
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = P.one[i];
    return r;
}

// 32-byte element <-> two 16-byte vector accesses
__device__ __forceinline__ Fr fr_load(const void *ptr)
{
    const uint4 *q = reinterpret_cast<const uint4 *>(ptr);
    uint4 lo = q[0], hi = q[1];
    Fr r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}

__device__ __forceinline__ void fr_store(void *ptr, const Fr &a)
{
    uint4 *q = reinterpret_cast<uint4 *>(ptr);
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}

// streaming store (`nt`): for data that crosses the chip once and must not push the twiddle tables out of L2
__device__ __forceinline__ void fr_store_nt(void *ptr, const Fr &a)
{
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    v4u *q = reinterpret_cast<v4u *>(ptr);
    v4u lo = {a.v[0], a.v[1], a.v[2], a.v[3]}, hi = {a.v[4], a.v[5], a.v[6], a.v[7]};
    __builtin_nontemporal_store(lo, q);
    __builtin_nontemporal_store(hi, q + 1);
}

__device__ __forceinline__ bool fr_is_zero(const Fr &a)
{
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
}

// r = a - p if a >= p else a   (a < 2p, possibly with an extra carry bit `hi`)
__device__ __forceinline__ Fr fr_reduce_once(const Fr &a, uint32_t hi, const FrParams &P)
{
    Fr d;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t t = (uint64_t)a.v[i] - P.p[i] - borrow;
        d.v[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
    // keep the difference when no final borrow, or when the carry bit covers it
    bool use_d = (borrow == 0) || (hi != 0);
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = use_d ? d.v[i] : a.v[i];
    return r;
}

__device__ __forceinline__ Fr fr_add(const Fr &a, const Fr &b, const FrParams &P)
{
    Fr s;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t t = (uint64_t)a.v[i] + b.v[i] + c;
        s.v[i] = (uint32_t)t;
        c = t >> 32;
    }
    return fr_reduce_once(s, (uint32_t)c, P);
}

__device__ __forceinline__ Fr fr_sub(const Fr &a, const Fr &b, const FrParams &P)
{
    Fr d;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t t = (uint64_t)a.v[i] - b.v[i] - borrow;
        d.v[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
    // add p back when the subtraction wrapped
    uint32_t mask = borrow ? 0xffffffffu : 0u;
    uint64_t c = 0;
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t t = (uint64_t)d.v[i] + (P.p[i] & mask) + c;
        r.v[i] = (uint32_t)t;
        c = t >> 32;
    }
    return r;
}

__device__ __forceinline__ Fr fr_neg(const Fr &a, const FrParams &P)
{
    Fr z = fr_zero();
    return fr_sub(z, a, P);
}

// a / 2 mod p (exact halving: add p when odd, shift right) — no multiplication
__device__ __forceinline__ Fr fr_halve(const Fr &a, const FrParams &P)
{
    uint32_t mask = (a.v[0] & 1) ? 0xffffffffu : 0u;
    uint32_t t[9];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t s = (uint64_t)a.v[i] + (P.p[i] & mask) + c;
        t[i] = (uint32_t)s;
        c = s >> 32;
    }
    t[8] = (uint32_t)c;
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = (t[i] >> 1) | (t[i + 1] << 31);
    return r;
}

// acc (64-bit column accumulator) += a * b;  cnt += carry-out.
// One v_mad_u64_u32 (32x32 + 64-bit addend, carry-out to an SGPR pair) plus one v_addc that folds the
// carry into a 32-bit overflow counter: two VALU instructions per limb product and no VCC-serialised
// carry chain.  The `_vs` form takes the second factor from an SGPR (the modulus limbs).
__device__ __forceinline__ void mac_vv(uint64_t &acc, uint32_t &cnt, uint32_t a, uint32_t b)
{
    uint64_t c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(c) : "v"(a), "v"(b));
    asm("v_addc_co_u32_e64 %0, %1, %0, 0, %1" : "+v"(cnt), "+s"(c));
}
__device__ __forceinline__ void mac_vs(uint64_t &acc, uint32_t &cnt, uint32_t a, uint32_t b_sgpr)
{
    uint64_t c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(c) : "v"(a), "s"(b_sgpr));
    asm("v_addc_co_u32_e64 %0, %1, %0, 0, %1" : "+v"(cnt), "+s"(c));
}

// Montgomery product a*b*R^-1 mod p, canonical output.
// Finely-integrated product scanning (FIPS): column k sums a_j*b_(k-j) and m_j*p_(k-j) into a
// 64-bit accumulator + overflow counter, derives m_k = lo32(acc) * (-p^-1) so that the column's low
// word cancels, then shifts the accumulator down one limb.  128 v_mad_u64_u32 + 128 v_addc per
// product; v_mad_u64_u32 issues at half the VALU rate on gfx950 (measured, bench/microbench.hip).
__device__ __forceinline__ Fr fr_mul(const Fr &a, const Fr &b, const FrParams &P)
{
    uint32_t m[8], t[8];
    uint64_t acc = 0;
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
        for (int j = 0; j <= k; j++) mac_vv(acc, cnt, a.v[j], b.v[k - j]);
#pragma unroll
        for (int j = 0; j < k; j++) mac_vs(acc, cnt, m[j], P.p[k - j]);
        m[k] = (uint32_t)acc * P.pinv;
        mac_vs(acc, cnt, m[k], P.p[0]);
        acc = (acc >> 32) | ((uint64_t)cnt << 32);
        cnt = 0;
    }
#pragma unroll
    for (int k = 8; k < 16; k++) {
#pragma unroll
        for (int j = k - 7; j < 8; j++) mac_vv(acc, cnt, a.v[j], b.v[k - j]);
#pragma unroll
        for (int j = k - 7; j < 8; j++) mac_vs(acc, cnt, m[j], P.p[k - j]);
        t[k - 8] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)cnt << 32);
        cnt = 0;
    }
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    return fr_reduce_once(r, (uint32_t)acc, P);
}

// Reference formulation (CIOS in plain C++, compiler-scheduled) kept for A/B measurements.
__device__ __forceinline__ Fr fr_mul_cios(const Fr &a, const Fr &b, const FrParams &P)
{
    uint32_t t[10];
#pragma unroll
    for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint64_t s = (uint64_t)a.v[j] * b.v[i] + ((uint64_t)t[j] + c);
            t[j] = (uint32_t)s;
            c = s >> 32;
        }
        uint64_t s = (uint64_t)t[8] + c;
        t[8] = (uint32_t)s;
        t[9] = (uint32_t)(s >> 32);

        uint32_t m = t[0] * P.pinv;
        s = (uint64_t)m * P.p[0] + t[0];
        c = s >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            s = (uint64_t)m * P.p[j] + ((uint64_t)t[j] + c);
            t[j - 1] = (uint32_t)s;
            c = s >> 32;
        }
        s = (uint64_t)t[8] + c;
        t[7] = (uint32_t)s;
        t[8] = t[9] + (uint32_t)(s >> 32);
    }
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    return fr_reduce_once(r, t[8], P);
}

__device__ __forceinline__ Fr fr_sqr(const Fr &a, const FrParams &P) { return fr_mul(a, a, P); }

// base^e for a 64-bit exponent (square-and-multiply, MSB first)
__device__ inline Fr fr_pow(const Fr &base, uint64_t e, const FrParams &P)
{
    Fr r = fr_one(P);
    bool started = false;
    for (int i = 63; i >= 0; i--) {
        if (started) r = fr_sqr(r, P);
        if ((e >> i) & 1) {
            r = fr_mul(r, base, P);
            started = true;
        }
    }
    return r;
}

}  // namespace hodor
