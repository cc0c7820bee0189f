"""GPU parity of the HANDLE API (-m gpu): device-resident `Polynomial` / `IopTree` / `FRIProofPrototype` objects
(hodor_poly_*_h, hodor_iop_*_h, hodor_fri_*_h — include/hodor_gpu.h) against the CPU oracle on the same seeded inputs,
bit for bit, method by method of src/polynomials/mod.rs, src/iop/mod.rs:79-92 and src/fri/mod.rs:43-54.  The C++ twin
(tests/host_cpp/test_host.cpp) replays the reference's own differential tests on the same entry points; this file pins
them to the oracle.  Nothing here reads /root/reference."""
import numpy as np
import pytest

import hodor_amd
from hodor_amd.handles import (COEFFICIENTS, VALUES, FriPrototypeHandle, IopTree, Polynomial, host_round_trips,
                               host_traffic, reset_host_round_trips)
from oracle import pyref as P
from oracle.oracle import array_to_ints, ints_to_array

pytestmark = pytest.mark.gpu

PYF = {"bn256": P.BN256, "experiments": P.EXPERIMENTS, "bn254": P.BN254}


def _int(row):
    return sum(int(row[i]) << (64 * i) for i in range(4))


# ---------------------------------------------------------------- construction, metadata, host access
@pytest.mark.parametrize("length", [1, 2, 3, 5, 8, 100, 1000, 4097])
def test_from_coeffs_pads_and_caches_the_domain(gpu_ctxs, oracles, field_name, length):
    """from_coeffs / from_values (src/polynomials/mod.rs:146-166, :722-742): zero-padded to the next power of two,
    exp / omega / omegainv / geninv / minv of that domain."""
    ctx, O = gpu_ctxs[field_name], oracles[field_name]
    a = O.random_elements(length, 300 + length)
    for form, make in ((COEFFICIENTS, Polynomial.from_coeffs), (VALUES, Polynomial.from_values)):
        p = make(ctx, a)
        n, k, omega = O.domain(length)
        assert p.size() == n and p.form == form
        info = p.info()
        assert info["exp"] == k and info["omega"] == omega
        assert info["omegainv"] == O.inverse(omega)
        assert info["geninv"] == O.inverse(ctx.generator)
        assert info["minv"] == O.inverse(O.from_canonical(n))
        host = p.as_ref()
        assert np.array_equal(host[:length], a) and not host[length:].any()
        assert np.array_equal(p.read(0, min(3, n)), host[:min(3, n)])
        p.free()


def test_beyond_the_two_adicity_is_an_error(gpu_ctxs):
    """Domain::new_for_size's Err (src/domains/mod.rs:30-32): bn254's scalar field has S = 28; refused before anything is
    allocated"""
    ctx = gpu_ctxs["bn254"]
    assert ctx.S == 28
    _, live0 = ctx.pool_stats()
    with pytest.raises(hodor_amd.HodorError) as e:
        Polynomial.new_for_size(ctx, COEFFICIENTS, (1 << 28) + 1)
    assert e.value.code == hodor_amd.ERR_SIZE
    p = Polynomial.new_for_size(ctx, COEFFICIENTS, 1 << 10)
    for call in (lambda: p.lde(1 << 19), lambda: p.coset_lde(1 << 19), lambda: Polynomial.lde_all([p, p], 1 << 19),
                 lambda: p.pad_by_factor(1 << 19), lambda: p.pad_to_size(1 << 29)):
        with pytest.raises(hodor_amd.HodorError) as e:
            call()
        assert e.value.code == hodor_amd.ERR_SIZE
    assert p.size() == 1 << 10
    p.free()
    assert ctx.pool_stats()[1] == live0


def test_read_write_elem_op_clone_equal(gpu_ctxs, oracles, field_name):
    ctx, O = gpu_ctxs[field_name], oracles[field_name]
    a = O.random_elements(256, 9)
    p = Polynomial.from_values(ctx, a)
    q = p.clone()
    assert p == q
    patch = O.random_elements(5, 10)
    p.write(17, patch)                                    # as_mut()[17..22] = patch
    exp = a.copy()
    exp[17:22] = patch
    assert not (p == q)
    c = _int(O.random_elements(1, 11)[0])
    p.elem_op(3, "add_constant", c)                       # as_mut()[3].add_assign(&c)
    p.elem_op(4, "sub_constant", c)
    p.elem_op(5, "scale", c)
    p.elem_op(6, "negate")
    p.elem_op(7, "square")
    p.elem_op(8, "pow", e=77)
    exp[3] = ints_to_array([O.add(_int(exp[3]), c)])[0]
    exp[4] = ints_to_array([O.sub(_int(exp[4]), c)])[0]
    exp[5] = ints_to_array([O.mul(_int(exp[5]), c)])[0]
    exp[6] = ints_to_array([O.sub(0, _int(exp[6]))])[0]
    exp[7] = ints_to_array([O.mul(_int(exp[7]), _int(exp[7]))])[0]
    exp[8] = ints_to_array([O.pow(_int(exp[8]), 77)])[0]
    assert np.array_equal(p.as_ref(), exp)
    assert np.array_equal(q.as_ref(), a)                  # the clone owns its vector
    assert np.array_equal(p.read(15, 10), exp[15:25])
    other_form = Polynomial.from_coeffs(ctx, a)
    assert not (other_form == q)                          # derive(PartialEq) compares the type too
    for x in (p, q, other_form):
        x.free()


@pytest.mark.parametrize("n", [2, 4, 64, 1 << 10, 1 << 16])
def test_as_mut_is_the_vector_until_it_is_written_back(gpu_ctxs, oracles, field_name, n):
    """Polynomial::as_mut() (src/polynomials/mod.rs:46) for the WHOLE slice, the way ALI's divisor precompute uses it
    (src/ali/per_register/mod.rs:112-160: new_for_size -> as_mut().chunks_mut() fill -> batch_inversion ->
    as_mut().chunks_mut() again): the host image is the vector while the borrow is open, one upload writes it back
    (explicitly, or before the next device operation), and a polynomial that is still new_for_size's zeros is not
    downloaded at all."""
    ctx, O = gpu_ctxs[field_name], oracles[field_name]
    a = O.random_elements(n, 21)
    a[a.sum(axis=1) == 0] = 1                              # (no zero element: the vector is inverted below)
    p = Polynomial.new_for_size(ctx, VALUES, n)
    reset_host_round_trips(ctx)
    with p.mutable() as m:                                 # zeros: nothing comes down
        assert m.shape == (n, 4) and not m.any()
        m[:] = a
        assert np.array_equal(p.as_ref(), a)               # as_ref() during the borrow reads the image
        assert np.array_equal(p.read(1, 1), a[1:2])
    up, down = host_traffic(ctx)
    assert down == 0 and host_round_trips(ctx) == 0
    assert up == (n * 32 if n > 4 else 0)                  # <= 4 elements travel as kernel arguments
    p.batch_inversion()                                    # (its zero check is a round trip of its own)
    exp = a.copy()
    O.poly_batch_inversion(exp)
    reset_host_round_trips(ctx)
    m = p.as_mut()                                         # one download of the inverted vector
    assert np.array_equal(m, exp)
    assert host_round_trips(ctx) == 1 and host_traffic(ctx) == (0, n * 32)
    m[0] = a[0]
    p.write(1, a[1:2])                                     # as_mut()[1] = v while the borrow is open: into the image
    exp[0], exp[1] = a[0], a[1]
    q = p.clone()                                          # NO commit_mut(): the next device operation writes back first
    assert np.array_equal(q.as_ref(), exp) and np.array_equal(p.as_ref(), exp)
    assert p == q
    p.square()                                             # ... and the image is stale after a device operation
    O.poly_unary(exp, "square")
    assert np.array_equal(p.as_ref(), exp)
    with p.mutable() as m:
        m[n - 1] = a[n - 1]
    exp[n - 1] = a[n - 1]
    tree_ok = n < 2 or IopTree.create(p).get_root() == bytes(O.iop_create(exp)[1])
    assert tree_ok
    assert np.array_equal(p.as_ref(), exp)
    p.free()
    q.free()


def test_as_mut_of_a_coefficient_polynomial_feeds_the_transform(gpu_ctxs, oracles):
    """q_poly.as_mut()[1] = F::one(); q_poly.as_mut()[0].sub_assign(&root) (src/ali/per_register/mod.rs:199-202) and a
    whole coefficient vector written on the host, then lde / fft straight after — the write-back is implicit."""
    ctx, O = gpu_ctxs["bn256"], oracles["bn256"]
    n = 1 << 12
    a = O.random_elements(n, 22)
    p = Polynomial.new_for_size(ctx, COEFFICIENTS, n)
    p.as_mut()[:] = a
    lde = p.lde(4)
    assert np.array_equal(lde.as_ref(), O.poly_lde(a, 4))
    m = p.as_mut()
    m[::2] = 0
    exp = a.copy()
    exp[::2] = 0
    p.fft()
    O.poly_fft(exp)
    assert p.form == VALUES and np.array_equal(p.as_ref(), exp)
    q = Polynomial.new_for_size(ctx, COEFFICIENTS, 2)
    one = O.random_elements(1, 23)
    q.as_mut()[1] = one[0]
    q.as_mut()[0] = one[0]
    assert np.array_equal(q.as_ref(), np.stack([one[0], one[0]]))
    d = q.evaluate_at(_int(one[0]))
    assert d == O.add(_int(one[0]), O.mul(_int(one[0]), _int(one[0])))
    for x in (p, q, lde):
        x.free()


# ---------------------------------------------------------------- transforms
@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 9, 10, 13, 16])
def test_transforms_match_oracle(gpu_ctxs, oracles, field_name, log_n):
    """fft / coset_fft / ifft / icoset_fft (+ _for_generator) — src/polynomials/mod.rs:611-638, :773-815"""
    ctx, O = gpu_ctxs[field_name], oracles[field_name]
    n = 1 << log_n
    a = O.random_elements(n, 40 + log_n)
    gen = _int(O.random_elements(1, 41)[0])
    cases = [
        ("fft", COEFFICIENTS, lambda p: p.fft(), lambda x: O.poly_fft(x)),
        ("coset_fft", COEFFICIENTS, lambda p: p.coset_fft(), lambda x: O.poly_coset_fft(x)),
        ("coset_fft_for_generator", COEFFICIENTS, lambda p: p.coset_fft_for_generator(gen),
         lambda x: O.poly_coset_fft_for_generator(x, gen)),
        ("ifft", VALUES, lambda p: p.ifft(), lambda x: O.poly_ifft(x)),
        ("icoset_fft", VALUES, lambda p: p.icoset_fft(), lambda x: O.poly_icoset_fft(x)),
        ("icoset_fft_for_generator", VALUES, lambda p: p.icoset_fft_for_generator(O.inverse(gen)),
         lambda x: O.poly_icoset_fft_for_generator(x, O.inverse(gen))),
    ]
    for name, form, dev, ref in cases:
        p = Polynomial._from_host(ctx, form, a)
        dev(p)
        exp = a.copy()
        ref(exp)
        assert p.form == 1 - form, name                       # the value changed its type
        assert np.array_equal(p.as_ref(), exp), name
        with pytest.raises(hodor_amd.HodorError) as e:        # ...and the old type's method is now a type error
            dev(p)
        assert e.value.code == hodor_amd.ERR_INVALID, name
        p.free()


@pytest.mark.parametrize("log_n,factor", [(0, 2), (3, 4), (8, 8), (10, 16), (12, 8), (14, 2)])
def test_lde_matches_oracle(gpu_ctxs, oracles, log_n, factor):
    """lde / coset_lde (src/polynomials/mod.rs:343-349 -> :418-482, :544-609) and the batch of src/prover/mod.rs:73-80"""
    ctx, O = gpu_ctxs["bn256"], oracles["bn256"]
    n = 1 << log_n
    regs = [O.random_elements(n, 60 + r) for r in range(3)]
    polys = [Polynomial.from_coeffs(ctx, r) for r in regs]
    for coset in (False, True):
        exp = [O.poly_lde(r, factor, coset=coset) for r in regs]
        one = polys[0].lde(factor, coset=coset)
        assert one.form == VALUES and one.size() == n * factor
        assert np.array_equal(one.as_ref(), exp[0])
        outs = Polynomial.lde_all(polys, factor, coset=coset)
        for o, e in zip(outs, exp):
            assert np.array_equal(o.as_ref(), e)
        for o in outs + [one]:
            o.free()
    assert np.array_equal(polys[0].as_ref(), regs[0])          # the input is left as it was
    for p in polys:
        p.free()


def test_lde_on_the_other_fields(gpu_ctxs, oracles, field_name):
    ctx, O = gpu_ctxs[field_name], oracles[field_name]
    a = O.random_elements(100, 3)                               # padded to 128 coefficients
    p = Polynomial.from_coeffs(ctx, a)
    padded = np.zeros((128, 4), dtype=np.uint64)
    padded[:100] = a
    for coset in (False, True):
        out = p.lde(16, coset=coset)
        assert np.array_equal(out.as_ref(), O.poly_lde(padded, 16, coset=coset))
        out.free()
    with pytest.raises(hodor_amd.HodorError):                   # assert!(factor.is_power_of_two()) :434
        p.lde(12)
    p.free()


def test_from_roots_and_the_other_spellings_of_lde(gpu_ctxs, oracles, field_name):
    """Polynomial::from_roots (:168-227) against the product computed with big integers; filtering_lde and the
    *_using_multiple_cosets(_naive) spellings return what lde / coset_lde return (:1030); into_coeffs consumes"""
    ctx, O, F = gpu_ctxs[field_name], oracles[field_name], PYF[field_name]
    for count in (1, 3, 13):
        roots = array_to_ints(O.random_elements(count, 900 + count))
        want = [1]                                                        # canonical coefficients of prod (x - r)
        for r in (F.from_mont(m) for m in roots):
            nxt = [0] * (len(want) + 1)
            for i, c in enumerate(want):
                nxt[i + 1] = (nxt[i + 1] + c) % F.p
                nxt[i] = (nxt[i] - c * r) % F.p
            want = nxt
        p = Polynomial.from_roots(ctx, roots)
        size = O.domain(count + 1)[0]
        assert p.form == COEFFICIENTS and p.size() == size
        got = [F.from_mont(m) for m in array_to_ints(p.as_ref())]
        assert got == want + [0] * (size - len(want))
        z = roots[0]
        assert p.evaluate_at(z) == 0                                      # a root is a root
        p.free()
    a = O.random_elements(64, 77)
    p = Polynomial.from_coeffs(ctx, a)
    base, cbase = p.lde(4), p.coset_lde(4)
    for name in ("filtering_lde", "lde_using_multiple_cosets", "lde_using_multiple_cosets_naive"):
        other = getattr(p, name)(4)
        assert other == base, name
        other.free()
    for name in ("coset_filtering_lde", "coset_lde_using_multiple_cosets", "coset_lde_using_multiple_cosets_naive"):
        other = getattr(p, name)(4)
        assert other == cbase, name
        other.free()
    assert np.array_equal(base.as_ref(), O.poly_lde(a, 4)) and np.array_equal(cbase.as_ref(), O.poly_lde(a, 4, coset=True))
    base.free()
    cbase.free()
    _, live0 = ctx.pool_stats()
    assert np.array_equal(p.into_coeffs(), a) and p.h is None
    assert ctx.pool_stats()[1] < live0


# ---------------------------------------------------------------- generic and pointwise methods
def test_generic_methods_match_oracle(gpu_ctxs, oracles, field_name):
    """distribute_powers / scale / negate / pad_by_factor / pad_to_size / trim_to_degree (:54-137)"""
    ctx, O = gpu_ctxs[field_name], oracles[field_name]
    n = 512
    a = O.random_elements(n, 70)
    g = _int(O.random_elements(1, 71)[0])
    p = Polynomial.from_coeffs(ctx, a)
    exp = a.copy()
    p.distribute_powers(g)
    O.distribute_powers(exp, g)
    assert np.array_equal(p.as_ref(), exp)
    p.scale(g)
    O.poly_unary(exp, "scale", c=g)
    p.negate()
    O.poly_unary(exp, "negate")
    assert np.array_equal(p.as_ref(), exp)
    p.pad_by_factor(2)
    assert p.size() == 2 * n and p.info()["exp"] == 10
    host = p.as_ref()
    assert np.array_equal(host[:n], exp) and not host[n:].any()
    for bad in (lambda: p.pad_by_factor(3), lambda: p.pad_to_size(1000), lambda: p.pad_to_size(n)):
        with pytest.raises(hodor_amd.HodorError) as e:
            bad()
        assert e.value.code == hodor_amd.ERR_SIZE
    for no_op in (2 * n - 1, 2 * n, 1 << 40, (1 << 64) - 1):           # size <= degree + 1: nothing happens (:129-131)
        p.trim_to_degree(no_op)
    assert np.array_equal(p.as_ref(), host)
    p.trim_to_degree(100)                                       # coefficients above x^100 become zero (:127-137)
    host = p.as_ref()
    assert np.array_equal(host[:101], exp[:101]) and not host[101:].any()
    assert p.size() == 2 * n
    p.free()


@pytest.mark.parametrize("n", [1, 2, 64, 1000, 1 << 14])
def test_values_arithmetic_matches_oracle(gpu_ctxs, oracles, field_name, n):
    """add_assign / sub_assign / mul_assign / add_assign_scaled / pow / square / add_constant / batch_inversion
    on Values (:744-771, :817-954)"""
    ctx, O = gpu_ctxs[field_name], oracles[field_name]
    size = O.domain(n)[0]
    a, b = O.random_elements(size, 80 + n), O.random_elements(size, 81 + n)
    s = _int(O.random_elements(1, 82)[0])
    p, q = Polynomial.from_values(ctx, a), Polynomial.from_values(ctx, b)
    exp = a.copy()
    for name in ("add", "sub", "mul"):
        getattr(p, name + "_assign")(q)
        O.poly_binary(exp, b, name)
        assert np.array_equal(p.as_ref(), exp), name
    p.add_assign_scaled(q, s)
    O.poly_add_scaled(exp, b, s)
    p.square()
    O.poly_unary(exp, "square")
    p.pow(13)
    O.poly_unary(exp, "pow", e=13)
    p.add_constant(s)
    O.poly_unary(exp, "add_constant", c=s)
    assert np.array_equal(p.as_ref(), exp)
    if not any(_int(r) == 0 for r in exp):
        p.batch_inversion()
        O.poly_batch_inversion(exp)
        assert np.array_equal(p.as_ref(), exp)
    assert np.array_equal(q.as_ref(), b)
    p.free()
    q.free()


def test_batch_inversion_of_a_zero_is_an_error(gpu_ctxs, oracles):
    """batch_inversion returns Err on a zero entry (:889-954)"""
    ctx, O = gpu_ctxs["bn256"], oracles["bn256"]
    a = O.random_elements(128, 5)
    a[77] = 0
    p = Polynomial.from_values(ctx, a)
    with pytest.raises(hodor_amd.HodorError):
        p.batch_inversion()
    p.free()


def test_coefficient_arithmetic_with_a_shorter_operand(gpu_ctxs, oracles, field_name):
    """Coefficients: add_assign / sub_assign / add_assign_scaled accept a SHORTER other (:640-683), mul_assign does
    not exist for them; evaluate_at (:685-711)"""
    ctx, O = gpu_ctxs[field_name], oracles[field_name]
    a, b = O.random_elements(256, 90), O.random_elements(64, 91)
    s = _int(O.random_elements(1, 92)[0])
    p, q = Polynomial.from_coeffs(ctx, a), Polynomial.from_coeffs(ctx, b)
    exp = a.copy()
    head = exp[:64].copy()
    p.add_assign(q)
    O.poly_binary(head, b, "add")
    p.sub_assign(q)
    O.poly_binary(head, b, "sub")
    p.add_assign_scaled(q, s)
    O.poly_add_scaled(head, b, s)
    exp[:64] = head
    assert np.array_equal(p.as_ref(), exp)
    with pytest.raises(hodor_amd.HodorError):                  # assert!(self.len >= other.len)
        q.add_assign(p)
    with pytest.raises(hodor_amd.HodorError) as e:             # mul_assign is a Values method
        p.mul_assign(p.clone())
    assert e.value.code == hodor_amd.ERR_INVALID
    z = _int(O.random_elements(1, 93)[0])
    assert p.evaluate_at(z) == O.evaluate_at(exp, z)
    assert q.evaluate_at(z) == O.evaluate_at(b, z)
    p.free()
    q.free()


@pytest.mark.parametrize("log_n", [1, 6, 12])
def test_evaluate_at_large_and_degree_one(gpu_ctxs, oracles, log_n):
    """evaluate_at (:685-711) and (coset_)evaluate_at_domain_for_degree_one (:229-290)"""
    ctx, O = gpu_ctxs["bn256"], oracles["bn256"]
    n = 1 << log_n
    a = O.random_elements(n, 95)
    z, alpha, c = (_int(r) for r in O.random_elements(3, 96))
    p = Polynomial.from_coeffs(ctx, a)
    assert p.evaluate_at(z) == O.evaluate_at(a, z, cpus=4)
    p.free()
    for coset in (False, True):
        d = Polynomial.degree_one_on_domain(ctx, n, alpha, c, coset=coset)
        assert d.form == VALUES
        assert np.array_equal(d.as_ref(), O.poly_degree_one_on_domain(n, alpha, c, coset=coset))
        d.free()


def test_quotient_term_is_the_deep_step(gpu_ctxs, oracles):
    """acc += alpha (f - value) / (x - z): the loop body of src/ali/per_register/deep.rs:75-146 in one pass, against the
    same step spelled with the reference's method calls on the oracle"""
    ctx, O = gpu_ctxs["bn256"], oracles["bn256"]
    n = 1 << 10
    f = O.random_elements(n, 97)
    z, value, alpha = (_int(r) for r in O.random_elements(3, 98))
    one = ctx.one
    dinv = O.poly_degree_one_on_domain(n, one, O.sub(0, z), coset=True)       # x - z on the coset
    O.poly_batch_inversion(dinv)
    pf, pd = Polynomial.from_values(ctx, f), Polynomial.from_values(ctx, dinv)
    acc = Polynomial.new_for_size(ctx, VALUES, n)
    exp = np.zeros_like(f)
    for rnd in range(2):
        acc.quotient_term(pf, pd, value, alpha, accumulate=rnd > 0)
        t = f.copy()
        O.poly_unary(t, "sub_constant", c=value)
        O.poly_binary(t, dinv, "mul")
        O.poly_add_scaled(exp, t, alpha)
        assert np.array_equal(acc.as_ref(), exp)
    for x in (pf, pd, acc):
        x.free()


# ---------------------------------------------------------------- oracles (Merkle trees)
@pytest.mark.parametrize("log_n", [1, 2, 3, 7, 12, 15])
@pytest.mark.parametrize("combiner", [hodor_amd.TRIVIAL, hodor_amd.COSET2])
def test_iop_matches_oracle(gpu_ctxs, oracles, log_n, combiner):
    """IOP::create / get_root / query (src/iop/blake2s_trivial_iop.rs:282-339) on a device-resident polynomial"""
    ctx, O = gpu_ctxs["bn256"], oracles["bn256"]
    n = 1 << log_n
    v = O.random_elements(n, 120 + log_n)
    p = Polynomial.from_values(ctx, v)
    if combiner == hodor_amd.COSET2 and n < 4:                  # a COSET2 tree has n / 2 >= 2 leaves
        with pytest.raises(hodor_amd.HodorError):
            IopTree.create(p, combiner)
        p.free()
        return
    tree = IopTree.create(p, combiner)
    nodes = O.iop_create(v) if combiner == hodor_amd.TRIVIAL else O.iop_create_coset2(v)
    assert tree.size() == n
    assert tree.get_root() == bytes(nodes[1])
    assert np.array_equal(tree.nodes()[1:], np.asarray(nodes)[1:])
    for idx in sorted({0, 1, n // 2, n - 1, (5 * n) // 7}):
        vals, path = tree.query(idx, p)
        if combiner == hodor_amd.TRIVIAL:
            assert vals == [_int(v[idx])]
            assert path == [bytes(x) for x in O.iop_path(nodes, v, idx)]
            assert O.iop_verify(tree.get_root(), vals[0], path, idx)
        else:
            lo = idx % (n // 2)
            assert vals == [_int(v[lo]), _int(v[lo + n // 2])]
            assert path == [bytes(x) for x in O.iop_path_coset2(nodes, v, idx)]
            assert O.iop_verify_coset2(tree.get_root(), vals[0], vals[1], path, lo) and len(path) == log_n - 1
    tree.free()
    p.free()


def test_iop_batch_and_roots_behind_one_wait(gpu_ctxs, oracles):
    """every register's oracle in one call, all roots in one round trip (src/prover/mod.rs:77-79)"""
    ctx, O = gpu_ctxs["bn256"], oracles["bn256"]
    regs = [O.random_elements(1 << 11, 130 + r) for r in range(4)]
    polys = [Polynomial.from_values(ctx, r) for r in regs]
    for combiner, make in ((hodor_amd.TRIVIAL, O.iop_create), (hodor_amd.COSET2, O.iop_create_coset2)):
        trees = IopTree.create_all(polys, combiner)
        reset_host_round_trips(ctx)
        roots = IopTree.get_roots(trees)
        assert host_round_trips(ctx) == 1
        assert roots == [bytes(make(r)[1]) for r in regs]
        assert [t.get_root() for t in trees] == roots           # cached: no further device reads
        assert host_round_trips(ctx) == 1
        for t in trees:
            t.free()
    for p in polys:
        p.free()


# ---------------------------------------------------------------- FRI on handles
@pytest.mark.parametrize("log_deg,lde_factor,out_deg", [(4, 4, 1), (6, 8, 2), (10, 8, 1), (12, 16, 4)])
@pytest.mark.parametrize("combiner", [hodor_amd.TRIVIAL, hodor_amd.COSET2])
@pytest.mark.parametrize("through", [False, True])
def test_fri_commit_on_a_handle_matches_oracle(gpu_ctxs, oracles, log_deg, lde_factor, out_deg, combiner, through):
    """NaiveFriIop::proof_from_lde (src/fri/mod.rs:43-54 -> fri_on_values.rs:11-159) and
    proof_from_lde_through_coefficients (:156-248), the prototype field by field"""
    ctx, O = gpu_ctxs["bn256"], oracles["bn256"]
    coeffs = O.random_elements(1 << log_deg, 140 + log_deg)
    p = Polynomial.from_coeffs(ctx, coeffs)
    lde = p.lde(lde_factor)
    exp = O.fri_commit(O.poly_lde(coeffs, lde_factor), lde_factor, out_deg, combiner=combiner,
                       through_coefficients=through)
    got = FriPrototypeHandle(lde, lde_factor, out_deg, combiner=combiner, through_coefficients=through)
    assert got.proto.serialized == exp["serialized"]
    assert got.proto.roots == exp["roots"] and got.proto.challenges == exp["challenges"]
    assert np.array_equal(got.proto.final_coeffs, exp["final_coeffs"])
    l0 = got.commitment(-1)
    assert l0.get_root() == exp["roots"][0] and l0.size() == lde.size()
    l0.free()
    for i in range(got.proto.num_steps):
        vals = got.intermediate_values(i)
        assert vals.form == VALUES and np.array_equal(vals.as_ref(), exp["inter_values"][i])
        t = got.commitment(i)
        assert t.get_root() == exp["roots"][i + 1]
        if vals.size() >= 4:
            again = IopTree.create(vals, combiner)                  # the committed tree IS the tree of these values
            assert np.array_equal(again.nodes()[1:], t.nodes()[1:])
            again.free()
        t.free()
        vals.free()
    n = lde.size()
    for idx in (1, 3, n // 2 + 1, n - 1):
        assert got.verify_prototype(idx)
    with pytest.raises(hodor_amd.HodorError) as e:                  # x^(n/2) == 1: "not in the LDE domain" (src/fri/verifier.rs:28-36)
        got.verify_prototype(2)
    assert e.value.code == hodor_amd.ERR_INVALID
    got.free()
    lde.free()
    p.free()


@pytest.mark.parametrize("combiner", [hodor_amd.TRIVIAL, hodor_amd.COSET2])
def test_several_fri_commits_at_once_are_the_separate_ones(gpu_ctxs, oracles, combiner):
    """hodor_fri_commit_batch_h — h1 and h2 of Prover::prove (src/prover/mod.rs:112-113) committed together, on streams of
    the context so that their small rounds overlap: the prototypes equal, byte for byte, the oracle's and the ones of one
    call each; ONE host round trip hands all of them over; the context's stream is ordered behind every commit (the
    queries right after see finished trees)."""
    ctx, O = gpu_ctxs["bn256"], oracles["bn256"]
    shapes = [(10, 8), (14, 8), (8, 8), (17, 8)]              # different sizes, as h1 (f domain) and h2 (g domain) are
    polys = [Polynomial.from_coeffs(ctx, O.random_elements(1 << lg, 170 + lg)) for lg, _ in shapes]
    ldes = [p.lde(f) for p, (_, f) in zip(polys, shapes)]
    exp = [O.fri_commit(O.poly_lde(p.as_ref(), f), f, 1, combiner=combiner) for p, (_, f) in zip(polys, shapes)]
    reset_host_round_trips(ctx)
    got = FriPrototypeHandle.commit_all(ldes, 8, 1, combiner=combiner)
    assert host_round_trips(ctx) == 1
    for g, e, l in zip(got, exp, ldes):
        assert g.proto.serialized == e["serialized"]
        assert g.verify_prototype(5)
        one = FriPrototypeHandle(l, 8, 1, combiner=combiner)
        assert one.proto.serialized == g.proto.serialized and one.produce_proof_bytes(5) == g.produce_proof_bytes(5)
        one.free()
    for x in got + ldes + polys:
        x.free()


@pytest.mark.parametrize("log_deg,lde_factor,index", [(3, 4, 7), (5, 4, 71), (5, 8, 255)])
def test_fri_proof_from_a_handle_matches_restated_query_producer(gpu_ctxs, log_deg, lde_factor, index):
    """produce_proof (src/fri/query_producer.rs:10-53) from device-resident values and trees: the bytes of the Python
    restatement, which the restated verifier accepts (src/fri/verifier.rs:131-289)"""
    F = P.BN256
    ctx = gpu_ctxs["bn256"]
    coeffs = [pow(7, 31 + i, F.p) for i in range(1 << log_deg)]
    lde = P.poly_lde(F, coeffs, lde_factor)
    proto = P.fri_commit(F, lde, lde_factor, 1)
    proof = P.fri_produce_proof(F, proto, lde, index, lde_factor, 1)
    assert P.fri_verify_proof_queries(F, proof, index, F.to_mont(lde[index]))
    p = Polynomial.from_coeffs(ctx, ints_to_array([F.to_mont(c) for c in coeffs]))
    d_lde = p.lde(lde_factor)
    assert array_to_ints(d_lde.as_ref()) == [F.to_mont(v) for v in lde]
    got = FriPrototypeHandle(d_lde, lde_factor, 1)
    raw = got.produce_proof_bytes(index)
    assert raw == P.fri_proof_to_bytes(proof)
    assert ctx.fri_verify_proof(raw, index, F.to_mont(lde[index])) is True
    got.free()
    d_lde.free()
    p.free()


# ---------------------------------------------------------------- residency
def test_a_commit_chain_crosses_pcie_once(gpu_ctxs, oracles):
    """coefficients -> lde -> oracle -> root: ONE device-to-host result (the root); pool blocks are reused"""
    ctx, O = gpu_ctxs["bn256"], oracles["bn256"]
    p = Polynomial.generated(ctx, COEFFICIENTS, 0, 1 << 16, 0x484F444F52)
    assert np.array_equal(p.read(5, 3), O.gen_elements(5, 3, 0x484F444F52))
    reset_host_round_trips(ctx)
    lde = p.lde(8)
    tree = IopTree.create(lde)
    root = tree.get_root()
    assert host_round_trips(ctx) == 1
    exp = O.iop_create(O.poly_lde(O.gen_elements(0, 1 << 16, 0x484F444F52), 8))
    assert root == bytes(exp[1])
    tree.free()
    lde.free()
    cached0, live0 = ctx.pool_stats()
    lde = p.lde(8)                                              # the same sizes again: served from the pool
    tree = IopTree.create(lde)
    assert tree.get_root() == root
    tree.free()
    lde.free()
    cached1, live1 = ctx.pool_stats()
    assert (cached1, live1) == (cached0, live0)
    p.free()


def test_handles_of_one_context_from_several_threads(gpu_ctxs, oracles):
    """The reference works on its registers from one scoped thread each (src/arp/per_register/mod.rs:43-49,
    src/polynomials/mod.rs:446-460): different handles of ONE context used from different threads at once — their work
    is serialised on the context's stream, the pool hands blocks from thread to thread — every result the oracle's."""
    import threading
    ctx, O = gpu_ctxs["bn256"], oracles["bn256"]
    n, factor, threads, rounds = 1 << 12, 8, 4, 6
    regs = [O.random_elements(n, 500 + t) for t in range(threads)]
    z = _int(O.random_elements(1, 499)[0])
    want = []
    for r in regs:
        lde = O.poly_lde(r, factor)
        sq = lde.copy()
        O.poly_unary(sq, "square")
        want.append((bytes(O.iop_create(lde)[1]), bytes(O.iop_create(sq)[1]), O.evaluate_at(r, z)))
    errors = []

    def work(t):
        try:
            for _ in range(rounds):
                p = Polynomial.from_coeffs(ctx, regs[t])
                lde = p.lde(factor)
                tree = IopTree.create(lde)
                sq = lde.clone()
                sq.square()
                tree2 = IopTree.create(sq)
                got = (tree.get_root(), tree2.get_root(), p.evaluate_at(z))
                if got != want[t]:
                    errors.append("thread %d: wrong result" % t)
                for x in (tree, tree2, sq, lde, p):
                    x.free()
        except Exception as exc:                                  # noqa: BLE001 — reported below
            errors.append("thread %d: %r" % (t, exc))

    ts = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_context_refuses_to_die_under_live_handles(oracles):
    """hodor_ctx_try_destroy: HODOR_ERR_INVALID while a handle of the context is alive"""
    import ctypes as C
    O = oracles["bn256"]
    ctx = hodor_amd.Context(P.BN256.p, P.BN256.g, device=0)
    p = Polynomial.from_values(ctx, O.random_elements(8, 1))
    assert ctx.L.hodor_ctx_try_destroy(C.c_void_p(ctx.h.value)) == hodor_amd.ERR_INVALID
    p.free()
    ctx.close()


def test_pool_keeps_the_block_just_released_whatever_the_cap():
    """The pool evicts the blocks idle for the longest time first and never the one just released (a prover repeating
    one shape must not pay hipMalloc + hipFree per call even when that shape alone is larger than the cache cap:
    HODOR_POOL_CACHE_GIB=0 here, in a process of its own because the knobs are read once)."""
    import os
    import subprocess
    import sys
    code = r"""
import numpy as np, hodor_amd
from hodor_amd.handles import Polynomial, VALUES
ctx = hodor_amd.Context(device=0)
ctx.trim()                                                 # (the start-up self-test leaves a few warm blocks behind)
assert ctx.pool_stats() == (0, 0)
a = Polynomial.new_for_size(ctx, VALUES, 1 << 16)          # 2 MiB
a.free()
cached, live = ctx.pool_stats()
assert (cached, live) == (2 << 20, 0), (cached, live)      # over the cap of 0 bytes, and kept
b = Polynomial.new_for_size(ctx, VALUES, 1 << 16)          # ...so that this one is served from the pool
assert ctx.pool_stats() == (0, 2 << 20)
c = Polynomial.new_for_size(ctx, VALUES, 1 << 18)          # 8 MiB
b.free()
c.free()                                                   # the older idle block goes, the one just released stays ...
assert ctx.pool_stats() == (10 << 20, 0), ctx.pool_stats() # ... but no free() hands memory back to HIP (hipFree drains the device):
ctx.synchronize()                                          # the evicted block waits for a call that waits anyway
assert ctx.pool_stats() == (8 << 20, 0), ctx.pool_stats()
ctx.close()
print("ok")
"""
    env = dict(os.environ, HODOR_POOL_CACHE_GIB="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
