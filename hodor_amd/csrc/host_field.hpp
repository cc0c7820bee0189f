// host_field.hpp — host-side (CPU) scalar arithmetic used by the C-ABI layer to derive constants
// (R, R^2, -p^-1, root of unity, domain generators, omega^-1, n^-1) and for the O(log n) host-side
// pieces of the IOP (path extraction, verification).  Scalar work only; every bulk operation runs
// in the HIP kernels.  Semantics: ff_ce `#[derive(PrimeField)]` for a 4-limb modulus
// (/root/reference/src/bn256.rs:4-7).
#pragma once
#include <stdint.h>
#include <string.h>

namespace hodor {

typedef unsigned __int128 u128_t;

struct HFr {
    uint64_t l[4];
    bool operator==(const HFr &o) const { return memcmp(l, o.l, 32) == 0; }
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
};

class HostField {
  public:
    uint64_t p[4];
    uint64_t pinv;
    HFr one, r2, generator, root_of_unity;
    uint32_t s, num_bits, capacity;

    static bool geq(const uint64_t a[4], const uint64_t b[4])
    {
        for (int i = 3; i >= 0; i--) {
            if (a[i] > b[i]) return true;
            if (a[i] < b[i]) return false;
        }
        return true;
    }
    static uint64_t add_n(uint64_t r[4], const uint64_t a[4], const uint64_t b[4])
    {
        uint64_t c = 0;
        for (int i = 0; i < 4; i++) {
            u128_t t = (u128_t)a[i] + b[i] + c;
            r[i] = (uint64_t)t;
            c = (uint64_t)(t >> 64);
        }
        return c;
    }
    static uint64_t sub_n(uint64_t r[4], const uint64_t a[4], const uint64_t b[4])
    {
        uint64_t bw = 0;
        for (int i = 0; i < 4; i++) {
            u128_t t = (u128_t)a[i] - b[i] - bw;
            r[i] = (uint64_t)t;
            bw = (uint64_t)(t >> 64) & 1;
        }
        return bw;
    }

    HFr add(const HFr &a, const HFr &b) const
    {
        HFr r;
        uint64_t c = add_n(r.l, a.l, b.l);
        if (c || geq(r.l, p)) sub_n(r.l, r.l, p);
        return r;
    }
    HFr sub(const HFr &a, const HFr &b) const
    {
        HFr r;
        if (sub_n(r.l, a.l, b.l)) add_n(r.l, r.l, p);
        return r;
    }
    HFr mul(const HFr &a, const HFr &b) const
    {
        // word-serial Montgomery multiplication, 64-bit words
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            uint64_t c = 0;
            for (int j = 0; j < 4; j++) {
                u128_t x = (u128_t)a.l[j] * b.l[i] + t[j] + c;
                t[j] = (uint64_t)x;
                c = (uint64_t)(x >> 64);
            }
            u128_t x = (u128_t)t[4] + c;
            t[4] = (uint64_t)x;
            t[5] = (uint64_t)(x >> 64);
            uint64_t m = t[0] * pinv;
            x = (u128_t)m * p[0] + t[0];
            c = (uint64_t)(x >> 64);
            for (int j = 1; j < 4; j++) {
                x = (u128_t)m * p[j] + t[j] + c;
                t[j - 1] = (uint64_t)x;
                c = (uint64_t)(x >> 64);
            }
            x = (u128_t)t[4] + c;
            t[3] = (uint64_t)x;
            t[4] = t[5] + (uint64_t)(x >> 64);
        }
        HFr r;
        memcpy(r.l, t, 32);
        if (t[4] || geq(r.l, p)) sub_n(r.l, r.l, p);
        return r;
    }
    HFr sqr(const HFr &a) const { return mul(a, a); }
    HFr pow(const HFr &a, uint64_t e) const
    {
        HFr r = one;
        bool started = false;
        for (int i = 63; i >= 0; i--) {
            if (started) r = sqr(r);
            if ((e >> i) & 1) { r = mul(r, a); started = true; }
        }
        return r;
    }
    HFr pow256(const HFr &a, const uint64_t e[4]) const
    {
        HFr r = one;
        for (int i = 255; i >= 0; i--) {
            r = sqr(r);
            if ((e[i >> 6] >> (i & 63)) & 1) r = mul(r, a);
        }
        return r;
    }
    bool inverse(const HFr &a, HFr *out) const
    {
        if (a.is_zero()) return false;
        uint64_t e[4], two[4] = {2, 0, 0, 0};
        sub_n(e, p, two);
        *out = pow256(a, e);   // Fermat: canonical result, same value as ff_ce's inverse()
        return true;
    }
    bool from_repr(const uint64_t c[4], HFr *out) const
    {
        if (geq(c, p)) return false;
        HFr t;
        memcpy(t.l, c, 32);
        *out = mul(t, r2);
        return true;
    }
    void into_repr(const HFr &a, uint64_t c[4]) const
    {
        HFr o = {{1, 0, 0, 0}};
        HFr t = mul(a, o);
        memcpy(c, t.l, 32);
    }
    HFr from_u64(uint64_t v) const
    {
        uint64_t c[4] = {v, 0, 0, 0};
        HFr r;
        from_repr(c, &r);
        return r;
    }

    // Domain::new_for_size, src/domains/mod.rs:21-44
    bool domain(uint64_t size, uint64_t *out_size, uint32_t *out_log, HFr *gen) const
    {
        uint64_t sz = 1;
        uint32_t k = 0;
        while (sz < size) { sz <<= 1; k++; if (k > 63) return false; }
        if (k > s) return false;
        HFr g = root_of_unity;
        for (uint32_t i = k; i < s; i++) g = sqr(g);
        *out_size = sz;
        *out_log = k;
        *gen = g;
        return true;
    }

    bool init(const uint64_t modulus[4], uint64_t gen)
    {
        memcpy(p, modulus, 32);
        if (!(p[0] & 1)) return false;
        if (p[3] == 0 || (p[3] >> 63)) return false;   // 4-limb field with 2p < 2^256
        uint64_t inv = 1;
        for (int i = 0; i < 63; i++) { inv *= inv; inv *= p[0]; }
        pinv = (uint64_t)0 - inv;
        int nb = 256;
        while (nb > 0 && !((p[(nb - 1) >> 6] >> ((nb - 1) & 63)) & 1)) nb--;
        num_bits = (uint32_t)nb;
        capacity = num_bits - 1;
        uint64_t x[4] = {1, 0, 0, 0};
        for (int i = 0; i < 512; i++) {
            uint64_t c = add_n(x, x, x);
            if (c || geq(x, p)) sub_n(x, x, p);
            if (i == 255) memcpy(one.l, x, 32);
        }
        memcpy(r2.l, x, 32);
        uint64_t t[4], o[4] = {1, 0, 0, 0};
        sub_n(t, p, o);
        s = 0;
        while (!(t[0] & 1)) {
            for (int i = 0; i < 3; i++) t[i] = (t[i] >> 1) | (t[i + 1] << 63);
            t[3] >>= 1;
            s++;
        }
        generator = from_u64(gen);
        root_of_unity = pow256(generator, t);
        return true;
    }
};

}  // namespace hodor
