"""An independent THIRD-PARTY implementation of the transform pins the oracle (and, on the GPU box, the library):
sympy's `ntt` / `intt` (sympy.discrete.transforms) compute A[k] = sum_i a[i] w^(ik) mod p over canonical residues with
w = g^((p - 1) / n), g the SMALLEST primitive root of p.  For both 4-limb fields of the reference that g is the field's
`multiplicative_generator` (7 for src/bn256.rs, 3 for src/experiments/mod.rs — sympy finds it by factoring p - 1), and
ff_ce's derive rule root_of_unity = g^t makes the domain generator of `Domain::new_for_size`
(/root/reference/src/domains/mod.rs:21-44) exactly that w.  So `Polynomial::fft` / `ifft`
(src/polynomials/mod.rs:611-624, :773-798) of canonical residues must equal sympy's output element for element —
code written by neither this repository nor the reference.  (It is a pin of the mathematics and of the root
convention, not a run of the Rust crate: DESIGN.md §4 keeps "parity unpinned" for that.)"""
import numpy as np
import pytest

sympy = pytest.importorskip("sympy")
from sympy.discrete.transforms import intt, ntt  # noqa: E402

from oracle import pyref as P  # noqa: E402
from oracle.oracle import array_to_ints, ints_to_array  # noqa: E402

FIELDS = {"bn256": P.BN256, "experiments": P.EXPERIMENTS}


def _canon(O, arr):
    return [O.to_canonical(x) for x in array_to_ints(arr)]


@pytest.mark.parametrize("name", ["bn256", "experiments"])
def test_smallest_primitive_root_is_the_reference_generator(name):
    F = FIELDS[name]
    assert sympy.primitive_root(F.p) == F.g


@pytest.mark.parametrize("name", ["bn256", "experiments"])
@pytest.mark.parametrize("log_n", [1, 2, 3, 5, 8, 10])
def test_oracle_fft_and_ifft_equal_sympy(oracles, name, log_n):
    O, F = oracles[name], FIELDS[name]
    n = 1 << log_n
    a = O.gen_elements(0, n, 0x53594D + log_n)
    can = _canon(O, a)
    want = [int(x) for x in ntt(can, F.p)]
    b = a.copy()
    O.poly_fft(b)                                  # the C restatement of Polynomial::fft (best_fft underneath)
    assert _canon(O, b) == want
    for variant in ("serial_fft", "serial_fft_radix_4"):
        if variant == "serial_fft_radix_4" and log_n % 2:
            continue
        c = a.copy()
        _, k, w = O.domain(n)
        getattr(O, variant)(c, w, k)
        assert _canon(O, c) == want, variant
    assert P.poly_fft(F, can) == want             # the Python big-int twin
    assert P.poly_ifft(F, want) == can
    # inverse: Polynomial::ifft = omega^-1 and the n^-1 scale
    back = a.copy()
    O.poly_fft(back)
    O.poly_ifft(back)
    assert np.array_equal(back, a)
    d = ints_to_array([O.from_canonical(v) for v in want])
    O.poly_ifft(d)
    assert _canon(O, d) == [int(x) for x in intt(want, F.p)] == can


@pytest.mark.parametrize("name", ["bn256", "experiments"])
@pytest.mark.parametrize("log_n,factor", [(3, 2), (5, 8), (7, 4)])
def test_oracle_coset_fft_and_lde_equal_sympy(oracles, name, log_n, factor):
    """coset_fft = fft of a[i] g^i (src/polynomials/mod.rs:626-631); lde / coset_lde = the transform of the zero-padded
    (and coset-scaled) coefficients on the n * factor domain (:343-349, asserted by the reference's own tests
    :1026-1031) — each against sympy's transform of the same list."""
    O, F = oracles[name], FIELDS[name]
    n = 1 << log_n
    a = O.gen_elements(0, n, 0x434F53 + log_n)
    can = _canon(O, a)
    scaled = [v * pow(F.g, i, F.p) % F.p for i, v in enumerate(can)]
    b = a.copy()
    O.poly_coset_fft(b)
    assert _canon(O, b) == [int(x) for x in ntt(scaled, F.p)]
    pad = [0] * (n * factor - n)
    assert _canon(O, O.poly_lde(a, factor)) == [int(x) for x in ntt(can + pad, F.p)]
    assert _canon(O, O.poly_lde(a, factor, coset=True)) == [int(x) for x in ntt(scaled + pad, F.p)]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["bn256", "experiments"])
@pytest.mark.parametrize("log_n", [4, 9, 11])
def test_gpu_fft_equals_sympy(gpu_ctxs, oracles, name, log_n):
    """The HIP path itself against the third-party transform (no oracle arithmetic in between: canonical residues are
    converted with the library's own from_repr / into_repr)."""
    ctx, O, F = gpu_ctxs[name], oracles[name], FIELDS[name]
    n = 1 << log_n
    can = _canon(O, O.gen_elements(0, n, 0x475055 + log_n))
    a = np.array([[(ctx.from_repr(v) >> (64 * i)) & (2**64 - 1) for i in range(4)] for v in can], dtype=np.uint64)
    ctx.poly_fft(a)
    got = [ctx.into_repr(sum(int(a[j][i]) << (64 * i) for i in range(4))) for j in range(n)]
    assert got == [int(x) for x in ntt(can, F.p)]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["bn256", "experiments"])
def test_gpu_lde_equals_sympy(gpu_ctxs, oracles, name):
    ctx, O, F = gpu_ctxs[name], oracles[name], FIELDS[name]
    n, factor = 1 << 7, 8
    can = _canon(O, O.gen_elements(0, n, 0x4C4445))
    a = np.array([[(ctx.from_repr(v) >> (64 * i)) & (2**64 - 1) for i in range(4)] for v in can], dtype=np.uint64)
    for coset in (False, True):
        out = ctx.poly_lde(a, factor, coset=coset)
        got = [ctx.into_repr(sum(int(out[j][i]) << (64 * i) for i in range(4))) for j in range(n * factor)]
        src = [v * pow(F.g, i, F.p) % F.p for i, v in enumerate(can)] if coset else can
        assert got == [int(x) for x in ntt(src + [0] * (n * factor - n), F.p)], coset


def _fri_rounds_against_sympy(F, coeffs_can, factor, out_deg, challenges_can, round_values_can):
    """A Merkle-free FRI invariant sympy can check on its own (src/fri/mod.rs:194-203 against
    src/fri/fri_on_values.rs:77-100): round k's vector must be the values, on the size n/2^k domain, of the polynomial whose
    coefficients are the previous ones folded a_(2i) + beta_k a_(2i+1) — i.e. sympy's intt of the round's vector must give
    exactly those folded coefficients followed by zeros (the low-degree property FRI is about)."""
    c = list(coeffs_can)
    for k, (beta, vec) in enumerate(zip(challenges_can, round_values_can)):
        c = [(c[2 * i] + beta * c[2 * i + 1]) % F.p for i in range(len(c) // 2)]
        got = [int(x) for x in intt(vec, F.p)]
        assert got[:len(c)] == c, ("round", k)
        assert not any(got[len(c):]), ("round", k, "degree")
        assert len(vec) == len(c) * factor
    return c[:out_deg]


@pytest.mark.parametrize("name", ["bn256", "experiments"])
@pytest.mark.parametrize("log_deg,factor,out_deg", [(3, 4, 1), (5, 8, 2), (6, 4, 1)])
def test_oracle_fri_folds_equal_sympy(oracles, name, log_deg, factor, out_deg):
    O, F = oracles[name], FIELDS[name]
    a = O.gen_elements(0, 1 << log_deg, 0x465249 + log_deg)
    can = _canon(O, a)
    lde = O.poly_lde(a, factor)
    assert _canon(O, lde) == [int(x) for x in ntt(can + [0] * (len(lde) - len(can)), F.p)]
    for combiner in (0, 1):              # the folds do not depend on the tree format, only the challenges do
        if combiner == 1 and factor * out_deg < 4:
            continue
        ref = O.fri_commit(lde, factor, out_deg, combiner=combiner)
        final = _fri_rounds_against_sympy(F, can, factor, out_deg, [O.to_canonical(b) for b in ref["challenges"]],
                                          [_canon(O, v) for v in ref["inter_values"]])
        assert _canon(O, ref["final_coeffs"]) == final


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["bn256", "experiments"])
@pytest.mark.parametrize("log_deg,factor,out_deg", [(4, 4, 1), (7, 8, 2), (10, 4, 1)])
def test_gpu_fri_folds_equal_sympy(gpu_ctxs, oracles, name, log_deg, factor, out_deg):
    """The device FRI commit (fold fused into the leaf launch, fused tail) against the same third-party invariant: the
    challenges are the library's own, every round's vector comes off the device, sympy does the rest."""
    ctx, O, F = gpu_ctxs[name], oracles[name], FIELDS[name]
    a = O.gen_elements(0, 1 << log_deg, 0x465250 + log_deg)
    can = _canon(O, a)
    lde = ctx.poly_lde(a, factor)
    n = len(lde)
    for combiner in (0, 1):
        proto = ctx.fri_commit(lde, factor, out_deg, combiner=combiner)
        rounds = [_canon(O, proto.intermediate_values(i, n >> (i + 1))) for i in range(proto.num_steps)]
        final = _fri_rounds_against_sympy(F, can, factor, out_deg, [ctx.into_repr(b) for b in proto.challenges], rounds)
        assert [ctx.into_repr(v) for v in array_to_ints(proto.final_coeffs)] == final
        proto.free()
