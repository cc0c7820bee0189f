"""Third-party published values that pin the oracle AND the library's host side independently of
anything written in this repository:

 * the field of src/bn256.rs is numerically the BLS12-381 scalar field; the constants below are the
   ones published in zkcrypto/bls12_381 `src/scalar.rs` (MODULUS, INV, R, R2, R3, GENERATOR = 7,
   S = 32, ROOT_OF_UNITY) — the same quantities ff_ce's derive computes for
   /root/reference/src/bn256.rs:4-7 (Montgomery R = 2^256, root_of_unity = generator^t);
 * BLAKE2s: RFC 7693 Appendix B ("abc") and the first keyed vectors of the BLAKE2 reference
   implementation's `testvectors/blake2s-kat.txt` (key = 00 01 .. 1f, input = 00 01 .. n-1).

 * the third modulus the tests drive the library with, the BN254 scalar field (not a field of the reference: it keeps
   "the modulus is a run-time parameter" honest): its multiplicative generator 5, two-adicity 28 and 2^28-th root of
   unity as published in arkworks `ark-bn254` (FrConfig::TWO_ADIC_ROOT_OF_UNITY; the same number is snarkjs's
   `Fr.nqr^t` root) — canonical residue; the derive rule root_of_unity = generator^t reproduces it.

They were typed from the published sources, not produced by code in this repository."""
import ctypes as C

import pytest

import hodor_amd
from oracle import pyref as P


def limbs(l):
    return sum(x << (64 * i) for i, x in enumerate(l))


MODULUS = limbs([0xffffffff00000001, 0x53bda402fffe5bfe, 0x3339d80809a1d805, 0x73eda753299d7d48])
INV = 0xfffffffeffffffff
R = limbs([0x00000001fffffffe, 0x5884b7fa00034802, 0x998c4fefecbc4ff5, 0x1824b159acc5056f])
R2 = limbs([0xc999e990f3f29c6d, 0x2b6cedcb87925c23, 0x05d314967254398f, 0x0748d9d99f59ff11])
R3 = limbs([0xc62c1807439b73af, 0x1b3e0d188cf06990, 0x73d13c71c7b5f418, 0x6e2a5bb9c8db33e9])
GENERATOR = limbs([0x0000000efffffff1, 0x17e363d300189c0f, 0xff9c57876f8457b0, 0x351332208fc5a8c4])
ROOT_OF_UNITY = limbs([0xb9b58d8c5f0e466a, 0x5b1b4c801819d7ec, 0x0af53ae352a31e64, 0x5bf3adda19e9b27b])
S = 32

BLAKE2S_KEYED_KAT = [   # blake2s-kat.txt, key = 000102..1f
    (0, "48a8997da407876b3d79c0d92325ad3b89cbb754d86ab71aee047ad345fd2c49"),
    (1, "40d15fee7c328830166ac3f918650f807e7e01e177258cdc0a39b11f598066f1"),
    (2, "6bb71300644cd3991b26ccd4d274acd1adeab8b1d7914546c1198bbe9fc9d803"),
]
BLAKE2S_ABC = "508c5e8c327c14e2e1a72ba34eeb452f37458b209ed63a294d999b4c86675982"   # RFC 7693 App. B


BN254_FR = 21888242871839275222246405745257275088548364400416034343698204186575808495617
BN254_FR_ROOT_OF_UNITY = 19103219067921713944291392827692070036145651957329286315305642004821462161904   # canonical, order 2^28


def test_bn254_scalar_field_constants(oracles):
    """A second, unrelated published field through the same generic code paths (ofield_init / HostField::init)."""
    O = oracles["bn254"]
    assert O.f.s == 28
    assert O.to_canonical(O.const("root_of_unity")) == BN254_FR_ROOT_OF_UNITY
    hodor_amd.build()
    ctx = hodor_amd.Context(BN254_FR, 5, device=-1)
    assert ctx.S == 28 and ctx.num_bits == 254
    assert ctx.into_repr(ctx.root_of_unity) == BN254_FR_ROOT_OF_UNITY
    assert ctx.into_repr(ctx.domain(1 << 28)[2]) == BN254_FR_ROOT_OF_UNITY
    ctx.close()


def test_experiments_field_is_the_stark_prime():
    """src/experiments/mod.rs:18-21 spells the modulus in decimal: it is StarkWare's published field prime
    2^251 + 17 * 2^192 + 1 (two-adicity 192, generator 3 as in the Cairo field), which fixes S and the limb count."""
    assert P.EXPERIMENTS.p == 2**251 + 17 * 2**192 + 1 == hodor_amd.EXPERIMENTS_FR_MODULUS
    assert P.EXPERIMENTS.g == 3 == hodor_amd.EXPERIMENTS_FR_GENERATOR
    t = (P.EXPERIMENTS.p - 1) >> 192
    assert t == 2**59 + 17 and t % 2 == 1


def test_reference_modulus_is_the_published_one():
    assert P.BN256.p == MODULUS == hodor_amd.BN256_FR_MODULUS and P.BN256.g == 7


def test_oracle_field_matches_published_constants(oracles):
    O = oracles["bn256"]
    assert O.f.pinv == INV and O.f.s == S
    assert O.one() == R and O.const("r2") == R2
    assert O.mul(R2, R2) == R3                              # mont(R2, R2) = R^3 mod p
    assert O.const("generator") == GENERATOR
    assert O.const("root_of_unity") == ROOT_OF_UNITY
    assert O.pow(ROOT_OF_UNITY, 1 << 32) == R and O.pow(ROOT_OF_UNITY, 1 << 31) != R
    assert O.domain(1 << 32)[2] == ROOT_OF_UNITY            # Domain::new_for_size at full 2-adicity


def test_library_host_field_matches_published_constants():
    hodor_amd.build()
    ctx = hodor_amd.Context(MODULUS, 7, device=-1)
    assert ctx.S == S and ctx.one == R and ctx.generator == GENERATOR and ctx.root_of_unity == ROOT_OF_UNITY
    assert ctx.from_repr(1) == R and ctx.mul(R2, 1) == R    # mont(R^2, 1) = R
    assert ctx.domain(1 << 32)[2] == ROOT_OF_UNITY
    ctx.close()


def test_oracle_blake2s_official_vectors(oracles):
    L = oracles["bn256"].L
    key = bytes(range(32))
    for n, exp in BLAKE2S_KEYED_KAT:
        out = (C.c_uint8 * 32)()
        L.o_blake2s(out, key, C.c_size_t(32), None, C.c_size_t(0), bytes(range(n)), C.c_size_t(n))
        assert bytes(out).hex() == exp, n
    out = (C.c_uint8 * 32)()
    L.o_blake2s(out, None, C.c_size_t(0), None, C.c_size_t(0), b"abc", C.c_size_t(3))
    assert bytes(out).hex() == BLAKE2S_ABC


def test_python_restatement_uses_the_same_blake2s():
    """hashlib (the generator of tests/golden/hodor_golden.json) reproduces the official vectors too."""
    import hashlib
    for n, exp in BLAKE2S_KEYED_KAT:
        assert hashlib.blake2s(bytes(range(n)), key=bytes(range(32))).hexdigest() == exp
    assert hashlib.blake2s(b"abc").hexdigest() == BLAKE2S_ABC
