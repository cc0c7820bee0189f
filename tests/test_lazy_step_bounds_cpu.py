"""Limb and accumulator bounds of the lazy hand-over between the radix-4 steps of k_ntt_pass (hodor_amd/csrc/ntt.hip,
fr9w3.cuh), replayed with Python integers that never wrap: a W3 step that receives un-carried sums (limbs < 5 * 2^29)
carries x0 in full and x1, x3 group-wise, multiplies, group-carries the mid-step sums, multiplies again and stores its own
sums un-carried.  Every 32-bit limb must stay below 2^32 and every 64-bit column accumulator below 2^64 for the WORST limbs
the bound allows (not only for random ones), the four results must be the butterfly's values mod p, and the stored limbs
must again be below 5 * 2^29 — the induction step of the bound stated in the kernel.  The constants (c5p, W3 entries) are
built the way csrc/abi.hip and k_pow_table_w3 build them.  Fields: the reference's two (src/bn256.rs:4-7,
src/experiments/mod.rs:18-21) and BN254's scalar field (1 mod 2^28: the generic Montgomery digit)."""
import random

import pytest

M29 = (1 << 29) - 1
U = 1 << 29

FIELDS = {
    "bn256": 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
    "stark252": 0x0800000000000011000000000000000000000000000000000000000000000001,
    "bn254": 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001,
}


def limbs_of(v, n=9):
    out = [(v >> (29 * i)) & M29 for i in range(n - 1)]
    out.append(v >> (29 * (n - 1)))
    return out


def value_of(l):
    return sum(x << (29 * i) for i, x in enumerate(l))


def spread(kp):
    """k*p with limb i < 8 raised by 2^29 borrowed from the limb above (abi.hip: c4p / spread_sum)."""
    l = limbs_of(kp)
    out = []
    for i, v in enumerate(l):
        if i < 8:
            v += U
        if i > 0:
            v -= 1
        out.append(v)
    assert value_of(out) == kp
    return out


def chk32(l):
    assert all(0 <= x < (1 << 32) for x in l), [hex(x) for x in l]
    return l


def normalize(a):
    a = list(a)
    for i in range(8):
        a[i + 1] += a[i] >> 29
        a[i] &= M29
    return chk32(a)


def normalize_groups(a):
    a = list(a)
    a[3] += a[2] >> 29
    a[2] &= M29
    a[6] += a[5] >> 29
    a[5] &= M29
    return chk32(a)


def add(a, b):
    return chk32([x + y for x, y in zip(a, b)])


def sub5(a, b, c5p):
    d = [c - y for c, y in zip(c5p, b)]
    assert all(x >= 0 for x in d)
    return chk32([x + y for x, y in zip(a, d)])


def w3_entry(w, p):
    return [limbs_of(w * pow(2, 87 * (c + 1), p) % p) for c in range(3)]


def mul3(a, W, p, pl, pinv):
    """fr9_mul3 (fr9w3.cuh) with an accumulator that must never pass 2^64."""
    m = [0, 0, 0]
    t = [0] * 9
    acc = 0
    for k in range(11):
        for c in range(3):
            for j in range(3):
                if 0 <= k - j < 9:
                    acc += a[3 * c + j] * W[c][k - j]
        for j in range(3):
            if j < k and k - j < 9:
                acc += m[j] * pl[k - j]
        if k < 3:
            m[k] = ((acc & 0xffffffff) * pinv) & M29
            acc += m[k] * pl[0]
            assert acc & M29 == 0
        else:
            t[k - 3] = acc & M29
        assert acc < (1 << 64), (k, hex(acc))
        acc >>= 29
    t[8] = acc
    assert acc < (1 << 32)
    return t


def worst_lazy(top_value, bound_units, rng=None):
    """limbs 0..7 at (or, with rng, randomly below) bound_units * 2^29 - 1, limb 8 so that the value is about top_value"""
    low = [bound_units * U - 1 if rng is None else rng.randrange(bound_units * U) for _ in range(8)]
    v8 = (top_value - value_of(low + [0])) >> 232
    assert v8 >= 0
    return low + [v8]


@pytest.mark.parametrize("name", sorted(FIELDS))
def test_lazy_w3_step_keeps_every_limb_and_column_in_range(name):
    p = FIELDS[name]
    pl = limbs_of(p)
    pinv = (-pow(p, -1, U)) % U
    c5p = spread(5 * p)
    rng = random.Random(2904)
    cases = [None] * 4 + [rng] * 60
    for r in cases:
        wa, wb, wbj = (rng.randrange(p) for _ in range(3))
        Wa, Wb, Wbj = w3_entry(wa, p), w3_entry(wb, p), w3_entry(wbj, p)
        # the value bound of a pass is 63p (ntt_launch_pass); the lazy limb bound of a stored sum is 5 * 2^29
        xs = [worst_lazy(rng.randrange(40 * p, 52 * p), 5, r) for _ in range(4)]
        vals = [value_of(x) for x in xs]
        x0, x1, x2, x3 = xs
        # ---- the W3 step as k_ntt_pass runs it with lazy_in and lazy_out
        x0 = normalize(x0)
        x1 = normalize_groups(x1)
        x3 = normalize_groups(x3)
        t = mul3(x1, Wa, p, pl, pinv)
        assert value_of(t) % p == vals[1] * wa % p and value_of(t) < 4 * p + (p >> 20) and max(t[:8]) < U
        x1 = sub5(x0, t, c5p)
        x0 = add(x0, t)
        t = mul3(x3, Wa, p, pl, pinv)
        x3 = sub5(x2, t, c5p)
        x2 = add(x2, t)
        assert max(x3) < 7 * U
        x2 = normalize_groups(x2)
        x3 = normalize_groups(x3)
        t = mul3(x2, Wb, p, pl, pinv)
        assert value_of(t) < 4 * p + (p >> 20)
        x2 = sub5(x0, t, c5p)
        x0 = add(x0, t)
        t = mul3(x3, Wbj, p, pl, pinv)
        assert value_of(t) < 4 * p + (p >> 20)
        x3 = sub5(x1, t, c5p)
        x1 = add(x1, t)
        # ---- results and the induction step of the limb bound
        a, b = (vals[0] + wa * vals[1]) % p, (vals[0] - wa * vals[1]) % p
        c, d = (vals[2] + wa * vals[3]) % p, (vals[2] - wa * vals[3]) % p
        assert value_of(x0) % p == (a + wb * c) % p
        assert value_of(x2) % p == (a - wb * c) % p
        assert value_of(x1) % p == (b + wbj * d) % p
        assert value_of(x3) % p == (b - wbj * d) % p
        for y in (x0, x1, x2, x3):
            assert max(y[:8]) < 5 * U


@pytest.mark.parametrize("name", sorted(FIELDS))
def test_normalized_step_output_is_the_same_value(name):
    """The last step of a pass (and a step in front of a W9 step) carries its sums: same values, limbs below 2^29."""
    p = FIELDS[name]
    rng = random.Random(7)
    for _ in range(20):
        x = worst_lazy(rng.randrange(40 * p, 62 * p), 5, rng)
        y = normalize(x)
        assert value_of(y) == value_of(x) and max(y[:8]) < U and y[8] < (1 << 29)


@pytest.mark.parametrize("name", sorted(FIELDS))
def test_offsets_leave_room_for_a_lazy_minuend(name):
    """A wave-uniform (W9) step that receives lazy sums carries x0, x1 and x3 in full and leaves x2 as it is: x2 only
    enters `x2 + t` and `x2 + (c11p - t)` before the mid-step carry.  The spread offsets are below 2^30 in limbs 0..7
    (abi.hip), so both stay below 7 * 2^29 < 2^32; its own outputs are a normalized x0 plus two such terms: < 5 * 2^29."""
    p = FIELDS[name]
    for k in (5, 11):
        c = spread(k * p)
        assert max(c[:8]) < (1 << 30) and min(c[:8]) >= U - 1
        assert 5 * U + max(c[:8]) < 7 * U
        assert U + 2 * max(c[:8]) <= 5 * U
