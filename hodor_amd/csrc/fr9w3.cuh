// fr9w3.cuh — K1'': data x table-constant products with a 3-limb Montgomery step ("W3 constants").
//
// Every multiplication of the transforms is (lazily reduced data) x (a constant that comes out of a
// table): butterfly twiddles (src/fft/radix4_fft/mod.rs:72-113), inter-pass twiddles, coset powers
// (src/fft/mod.rs:110-123), the n^-1 scale (src/polynomials/mod.rs:777-787).  A table can therefore
// hold the constant pre-shifted three ways,
//
//     W_c = w * 2^(87 (c + 1)) mod p,   c = 0, 1, 2        (w the plain integer value, 87 = 3 x 29)
//
// and with x split into its three 87-bit limb groups  x = X_0 + X_1 2^87 + X_2 2^174:
//
//     T = X_0 W_0 + X_1 W_1 + X_2 W_2  ==  x w 2^87  (mod p),      11 columns of 29 bits
//     r = (T + m p) / 2^87             ==  x w       (mod p),      m = -T p^-1 mod 2^87
//
// i.e. only THREE Montgomery steps instead of nine: 81 + 27 v_mad_u64_u32 and 3 v_mul_lo against the
// 162 + 9 of fr9_mul, for a table entry of 27 limbs instead of 9.  Data keeps whatever form it is in
// (the reference's R = 2^256 Montgomery image stays that image: the constant is a plain integer).
//
// Bounds (p < 2^255): with x normalized (limbs 0..7 < 2^29, value < 2^261) every X_c < 2^87, so
// T < 3 * 2^87 p and r < 4p, normalized.  Column sums: at most 9 data terms (< 2^29 * 2^29 when x is
// normalized; lazy limbs are tolerated as long as the nine of them sum below 60 * 2^29, see fr9_normalize_groups)
// + 3 reduction terms < 2^64.
#pragma once
#ifndef HODOR_HOST_TEST
#include "fr9.cuh"
#endif

namespace hodor {

struct Fr9W3 {
    uint32_t w[3][9];
};

#ifndef FR9_MAD
#define FR9_MAD(acc, x, y) do { acc += (uint64_t)(x) * (y); } while (0)
#endif

// P1: the modulus is 1 mod 2^29 (true for both fields of the reference, src/bn256.rs:5 and
// src/experiments/mod.rs:19: p = 1 mod 2^32 resp. 2^192), so -p^-1 = -1 mod 2^29 and the Montgomery
// quotient digit is m = -T mod 2^29: a subtraction instead of a (half-rate) v_mul_lo_u32.
template <bool P1>
__device__ __forceinline__ uint32_t fr9_mont_digit(uint32_t lo, const Fr9Params &P)
{
    return (P1 ? 0u - lo : lo * P.pinv) & HODOR_M29;
}

// group-wise carry propagation: enough for fr9_mul3, whose bound only needs every 87-bit GROUP below 2^87
// (limbs 2 and 5 carried into 3 and 6; the top group is bounded by the value itself).  For x with lazy limbs
// < 7 * 2^29 (what k_ntt_pass hands it: a stored sum < 5 * 2^29 plus one offset subtraction) and value < 2^261:
// X_0, X_1 < 2^87 (1 + 2^-26), X_2 < 2^87; a column of the product takes every data limb at most once, seven of them
// < 7 * 2^29 and limbs 2 and 5 < 2^29 after the carry, each times a table limb < 2^29: <= 51 * 2^58, + 3 * 2^58 of
// reduction terms + the carry < 2^64; the product stays < (4 + 2^-25) p — 6 instructions instead of 24
// (replayed with the worst limbs in tests/test_lazy_step_bounds_cpu.py).
__device__ __forceinline__ void fr9_normalize_groups(Fr9 &a)
{
    a.v[3] += a.v[2] >> 29;
    a.v[2] &= HODOR_M29;
    a.v[6] += a.v[5] >> 29;
    a.v[5] &= HODOR_M29;
}

template <bool P1 = false>
__device__ __forceinline__ Fr9 fr9_mul3(const Fr9 &a, const Fr9W3 &W, const Fr9Params &P)
{
    uint32_t m[3];
    Fr9 t;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 11; k++) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
#pragma unroll
            for (int j = 0; j < 3; j++) {
                if (k - j >= 0 && k - j < 9) FR9_MAD(acc, a.v[3 * c + j], W.w[c][k - j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            if (j < k && k - j < 9) FR9_MAD(acc, m[j], P.p[k - j]);
        }
        if (k < 3) {
            m[k] = fr9_mont_digit<P1>((uint32_t)acc, P);
            FR9_MAD(acc, m[k], P.p[0]);
        } else {
            t.v[k - 3] = (uint32_t)acc & HODOR_M29;
        }
        acc >>= 29;
    }
    t.v[8] = (uint32_t)acc;
    return t;
}

// two products by the same W3 constant with their accumulator chains interleaved and un-pinned: no mad depends on the
// one issued just before it (no wait state between them), at the price of two live accumulators and twelve more live
// limbs.  Experiment of round 4 (-DHODOR_EXP_X2, profiles/r04/diet_ab.txt): see DESIGN.md §8.
template <bool P1 = false>
__device__ __forceinline__ void fr9_mul3x2(Fr9 &a, Fr9 &b, const Fr9W3 &W, const Fr9Params &P)
{
    uint32_t ma[3], mb[3];
    Fr9 ta, tb;
    uint64_t acca = 0, accb = 0;
#pragma unroll
    for (int k = 0; k < 11; k++) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
#pragma unroll
            for (int j = 0; j < 3; j++) {
                if (k - j >= 0 && k - j < 9) {
                    acca += (uint64_t)a.v[3 * c + j] * W.w[c][k - j];
                    accb += (uint64_t)b.v[3 * c + j] * W.w[c][k - j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            if (j < k && k - j < 9) {
                acca += (uint64_t)ma[j] * P.p[k - j];
                accb += (uint64_t)mb[j] * P.p[k - j];
            }
        }
        if (k < 3) {
            ma[k] = fr9_mont_digit<P1>((uint32_t)acca, P);
            mb[k] = fr9_mont_digit<P1>((uint32_t)accb, P);
            acca += (uint64_t)ma[k] * P.p[0];
            accb += (uint64_t)mb[k] * P.p[0];
        } else {
            ta.v[k - 3] = (uint32_t)acca & HODOR_M29;
            tb.v[k - 3] = (uint32_t)accb & HODOR_M29;
        }
        acca >>= 29;
        accb >>= 29;
    }
    ta.v[8] = (uint32_t)acca;
    tb.v[8] = (uint32_t)accb;
    a = ta;
    b = tb;
}

// ---- W9: the same idea with one-limb groups, for WAVE-UNIFORM constants ------------------------------------
//     V_c = w * 2^(29 (c + 1)) mod p,  c = 0 .. 8;     T = sum_c x_c V_c  ==  x w 2^29 (mod p),  9 columns
//     r = (T + m p) / 2^29  ==  x w (mod p),           m = -T p^-1 mod 2^29
// ONE Montgomery step: 81 + 9 v_mad_u64_u32 and 1 v_mul_lo against the 108 + 3 of fr9_mul3.  The entry is 81
// limbs — far too much to fetch per lane, but a constant shared by the whole wave is read with scalar loads and
// enters the multiplier as an SGPR operand (free, like the modulus): the twiddles of the first radix-4 steps of
// a pass, which take only a handful of values (k_ntt_pass).  Table layout: column-major, V[k][c] = limb k of
// V_c, 9 words per column padded to 12 (so that a column is three aligned dwordx4), 108 words per entry.
// Bounds: x normalized and < 2^261 (every limb < 2^29) => T < 9 * 2^29 p, r < 10 p, normalized.
typedef const __attribute__((address_space(4))) uint32_t *W9Ptr;

template <bool P1 = false>
__device__ __forceinline__ Fr9 fr9_mul9(const Fr9 &a, W9Ptr V, const Fr9Params &P)
{
    Fr9 t;
    uint64_t acc = 0;
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int c = 0; c < 9; c++) FR9_MAD(acc, a.v[c], V[12 * k + c]);
        if (k == 0) m = fr9_mont_digit<P1>((uint32_t)acc, P);
        FR9_MAD(acc, m, P.p[k]);
        if (k > 0) t.v[k - 1] = (uint32_t)acc & HODOR_M29;
        acc >>= 29;
    }
    t.v[8] = (uint32_t)acc;
    return t;
}

// two products by the same constant, column by column: the scalar operands of a column are loaded once
template <bool P1 = false>
__device__ __forceinline__ void fr9_mul9x2(Fr9 &a, Fr9 &b, W9Ptr V, const Fr9Params &P)
{
    Fr9 ta, tb;
    uint64_t acca = 0, accb = 0;
    uint32_t ma = 0, mb = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int c = 0; c < 9; c++) {
            const uint32_t v = V[12 * k + c];
            FR9_MAD(acca, a.v[c], v);
            FR9_MAD(accb, b.v[c], v);
        }
        if (k == 0) {
            ma = fr9_mont_digit<P1>((uint32_t)acca, P);
            mb = fr9_mont_digit<P1>((uint32_t)accb, P);
        }
        FR9_MAD(acca, ma, P.p[k]);
        FR9_MAD(accb, mb, P.p[k]);
        if (k > 0) {
            ta.v[k - 1] = (uint32_t)acca & HODOR_M29;
            tb.v[k - 1] = (uint32_t)accb & HODOR_M29;
        }
        acca >>= 29;
        accb >>= 29;
    }
    ta.v[8] = (uint32_t)acca;
    tb.v[8] = (uint32_t)accb;
    a = ta;
    b = tb;
}

// 27 limbs stored as 7 x 16 bytes (28 words, the last one unused)
__device__ __forceinline__ Fr9W3 fr9w3_load(const void *ptr)
{
    const uint4 *q = reinterpret_cast<const uint4 *>(ptr);
    uint32_t f[28];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        uint4 v = q[i];
        f[4 * i] = v.x; f[4 * i + 1] = v.y; f[4 * i + 2] = v.z; f[4 * i + 3] = v.w;
    }
    Fr9W3 r;
#pragma unroll
    for (int i = 0; i < 27; i++) r.w[i / 9][i % 9] = f[i];
    return r;
}

__device__ __forceinline__ void fr9w3_store(void *ptr, const Fr9W3 &a)
{
    uint4 *q = reinterpret_cast<uint4 *>(ptr);
    uint32_t f[28];
#pragma unroll
    for (int i = 0; i < 27; i++) f[i] = a.w[i / 9][i % 9];
    f[27] = 0;
#pragma unroll
    for (int i = 0; i < 7; i++) q[i] = make_uint4(f[4 * i], f[4 * i + 1], f[4 * i + 2], f[4 * i + 3]);
}

}  // namespace hodor
