"""CPU tests: pin the C oracle (oracle/hodor_oracle.c) against
 (1) the committed golden vectors (tests/golden/hodor_golden.json — Python big-int + hashlib),
 (2) the constants of SURVEY.md Appendix A/B,
 (3) the reference's own differential identities, restated.
Parity status: unpinned against the Rust binary (see oracle/hodor_oracle.h)."""
import json
import os

import numpy as np
import pytest

from oracle import pyref as P
from oracle.oracle import array_to_ints, ints_to_array

PYF = {"bn256": P.BN256, "experiments": P.EXPERIMENTS, "bn254": P.BN254}
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "hodor_golden.json")))


def h2i(xs):
    return [int(x, 16) for x in xs]


def mont_array(F, canon):
    return ints_to_array([F.to_mont(v) for v in canon])


def canon_list(F, arr):
    return [F.from_mont(v) for v in array_to_ints(arr)]


def test_field_constants_appendix_a(oracles):
    O = oracles["bn256"]
    assert O.one() == 0x1824B159ACC5056F998C4FEFECBC4FF55884B7FA0003480200000001FFFFFFFE
    assert O.const("r2") == 0x0748D9D99F59FF1105D314967254398F2B6CEDCB87925C23C999E990F3F29C6D
    assert O.f.pinv == 0xFFFFFFFEFFFFFFFF and O.f.s == 32
    assert O.to_canonical(O.const("root_of_unity")) == \
        0x16A2A19EDFE81F20D09B681922C813B4B63683508C2280B93829971F439F0D2B
    assert O.to_canonical(O.domain(1 << 24)[2]) == \
        0x291CF6D68823E6876E0BCD91EE76273072CF6A8029B7D7BC92CF4DEB77BD779C
    assert O.to_canonical(O.domain(1 << 30)[2]) == \
        0x2C6D4E4511657E1E1339A815DA8B398FED3A181FABB30ADC694341F608C9DD56
    with pytest.raises(ValueError):
        O.domain(1 << 33)                         # S = 32
    E = oracles["experiments"]
    assert E.one() == 0x07fffffffffffdf0ffffffffffffffffffffffffffffffffffffffffffffffe1
    assert E.f.pinv == 0xFFFFFFFFFFFFFFFF and E.f.s == 192
    assert E.to_canonical(E.const("root_of_unity")) == \
        0x005282DB87529CFA3F0464519C8B0FA5AD187148E11A61616070024F42F8EF94


def test_blake2s_against_hashlib(oracles):
    import ctypes as C
    O = oracles["bn256"]
    for msg in (b"", b"abc", b"\x00" * 64, bytes(range(200))):
        out = (C.c_uint8 * 32)()
        O.L.o_blake2s(out, P.IOP_KEY, C.c_size_t(19), P.IOP_PERSONAL, C.c_size_t(7), msg, C.c_size_t(len(msg)))
        assert bytes(out) == P.b2s(msg)
    assert P.b2s(b"").hex() == GOLD["blake2s"]["h_empty"] == \
        "a61dd261a9b23522c19ebdecc9b5755882c1b4f3940d3437029d99120ab1b437"   # Appendix B


def test_make_small_tree_appendix_b(oracles):
    for name, root, chal in (
        ("bn256", "661512723ab4cfa09bdd1aad0e9f1cc69356055f99a9528b016f35b8c5fe706b",
         0x261512723AB4CFA09BDD1AAD0E9F1CC69356055F99A9528B016F35B8C5FE706B),
        ("experiments", "fdf489862b4402468d94f026c014e1ca0129f421a55ce5e1df838eab8eefbd22",
         2693624083161027199676761339081382127720247941633710928972358601781283896610)):
        O, F = oracles[name], PYF[name]
        nodes = O.iop_create(ints_to_array([F.R] * 16))
        assert bytes(nodes[1]).hex() == root == GOLD[name]["cases"]["make_small_tree"]["root"]
        assert O.to_canonical(O.interpret_hash(bytes(nodes[1]))) == chal


def test_oracle_vs_golden_transforms(oracles, field_name):
    O, F, cases = oracles[field_name], PYF[field_name], GOLD[field_name]["cases"]
    for key, c in cases.items():
        if key.startswith("ntt_"):
            a = mont_array(F, h2i(c["input"]))
            for name in ("fft", "ifft", "coset_fft", "icoset_fft"):
                b = a.copy()
                getattr(O, "poly_" + name)(b)
                assert canon_list(F, b) == h2i(c[name]), (key, name)
        elif key.startswith("lde_"):
            a = mont_array(F, h2i(c["input"]))
            assert canon_list(F, O.poly_lde(a, c["factor"])) == h2i(c["lde"])
            assert canon_list(F, O.poly_lde(a, c["factor"], coset=True)) == h2i(c["coset_lde"])


def test_oracle_vs_golden_merkle_and_fri(oracles, field_name):
    O, F, cases = oracles[field_name], PYF[field_name], GOLD[field_name]["cases"]
    for key, c in cases.items():
        if key.startswith("merkle_"):
            leafs = ints_to_array(h2i(c["leafs_mont"]))
            nodes = O.iop_create(leafs)
            assert [bytes(x).hex() for x in nodes[1:]] == c["nodes"][1:]
            assert O.to_canonical(O.interpret_hash(bytes(nodes[1]))) == int(c["challenge"], 16)
            n = len(leafs)
            assert [bytes(x).hex() for x in O.iop_path(nodes, leafs, 3 % n)] == c["path_3"]
        elif key.startswith("fri_"):
            coeffs = mont_array(F, h2i(c["coeffs"]))
            lde = O.poly_lde(coeffs, c["lde_factor"])
            r = O.fri_commit(lde, c["lde_factor"], c["out_deg_plus_one"])
            assert r["serialized"].hex() == c["serialized"]
            assert [x.hex() for x in r["roots"]] == c["roots"]
            assert [O.to_canonical(x) for x in r["challenges"]] == h2i(c["challenges"])


@pytest.mark.parametrize("log_n", [2, 4, 6, 10, 12])
def test_reference_differential_identities(oracles, field_name, log_n):
    """radix-2 == radix-4 == parallel == naive DFT (src/fft/mod.rs:66-184); thread-count independence
    (test_worker_size :281-328); multi-coset LDE == filtering LDE == FFT of padded
    (src/polynomials/mod.rs:988-1130)."""
    O = oracles[field_name]
    n = 1 << log_n
    a = O.random_elements(n, log_n)
    _, k, w = O.domain(n)
    r2 = a.copy(); O.serial_fft(r2, w, k)
    r4 = a.copy(); O.serial_fft_radix_4(r4, w, k)
    assert np.array_equal(r2, r4)
    for log_cpus in (1, 2):
        pf = a.copy(); O.parallel_fft(pf, w, k, log_cpus)
        assert np.array_equal(r2, pf)
    for cpus in (1, 3, 16):
        bf = a.copy(); O.best_fft(bf, w, k, cpus)
        assert np.array_equal(r2, bf)
    if log_n <= 6:
        assert np.array_equal(r2, O.naive_dft(a, w))
    inv = r2.copy(); O.poly_ifft(inv)
    assert np.array_equal(inv, a)
    for factor in (2, 8):
        lde = O.poly_lde(a, factor)
        pad = np.zeros((n * factor, 4), dtype=np.uint64); pad[:n] = a
        _, K, W = O.domain(n * factor)
        f1 = pad.copy(); O.serial_fft(f1, W, K)
        f2 = pad.copy(); O.serial_lde(f2, W, K, factor)
        assert np.array_equal(lde, f1) and np.array_equal(lde, f2)
        # coset LDE evaluates at g * Omega^idx
        clde = O.poly_lde(a, factor, coset=True)
        g = O.const("generator")
        for idx in (0, 1, n * factor - 1):
            pt = O.mul(g, O.pow(W, idx))
            assert array_to_ints(clde[idx:idx + 1])[0] == O.evaluate_at(a, pt)


@pytest.mark.parametrize("log_n", [1, 3, 6, 9, 12])
def test_dit_fft_three_way_and_pruning(oracles, field_name, log_n):
    """serial_DIT_fft / parallel_DIT_fft / best_DIT_fft (src/fft/dit_fft/mod.rs:4-123) == serial_fft, and
    test_fft_prunning (src/fft/mod.rs:187-230): with only the first n / 2^k inputs non-zero, the transform
    pruned by non_zero_entries_count equals the unpruned one — the schedule hodor_lde(nnz) replaces."""
    O = oracles[field_name]
    n = 1 << log_n
    _, k, w = O.domain(n)
    a = O.random_elements(n, 40 + log_n)
    ref = a.copy(); O.serial_fft(ref, w, k)
    d = a.copy(); O.serial_dit_fft(d, w, k, n)
    assert np.array_equal(d, ref)
    for log_cpus in (1, 2):
        if log_n >= log_cpus:
            pd = a.copy(); O.parallel_dit_fft(pd, w, k, log_cpus, n)
            assert np.array_equal(pd, ref)
    for cpus in (1, 4, 64):
        bd = a.copy(); O.best_dit_fft(bd, w, k, n, cpus=cpus)
        assert np.array_equal(bd, ref)
    for log_nz in range(0, log_n + 1):
        nz = 1 << log_nz
        z = a.copy(); z[nz:] = 0
        full = z.copy(); O.serial_fft(full, w, k)
        pr = z.copy(); O.serial_dit_fft(pr, w, k, nz)
        assert np.array_equal(pr, full), log_nz
        pp = z.copy(); O.best_dit_fft(pp, w, k, nz, cpus=4)
        assert np.array_equal(pp, full), log_nz
        if log_nz < log_n:                      # serial_lde (lde.rs) is the same zero-aware transform
            sl = z.copy(); O.serial_lde(sl, w, k, n // nz)
            assert np.array_equal(sl, full), log_nz


@pytest.mark.parametrize("log_n", [2, 4, 6, 8, 12])
def test_parallel_radix4_and_parallel_lde_restatements(oracles, field_name, log_n):
    """Round 6: the last two functions of SURVEY §8(a) without a restatement — parallel_fft_radix_4
    (src/fft/radix4_fft/mod.rs:125-184) with its best_fft (:5-20), parallel_lde / best_lde (src/fft/lde.rs:128-193, :4-13) —
    against the serial forms, against the big-int twins of oracle/pyref.py (which follow the Rust loops with Python
    integers) and against the reference's own asserts on log_n / log_cpus."""
    O, F = oracles[field_name], PYF[field_name]
    n = 1 << log_n
    a = O.random_elements(n, 60 + log_n)
    _, k, w = O.domain(n)
    ref = a.copy(); O.serial_fft(ref, w, k)
    for log_cpus in (0, 2, 4):
        if log_cpus <= log_n:
            r = a.copy(); O.parallel_fft_radix_4(r, w, k, log_cpus)
            assert np.array_equal(r, ref), log_cpus
    for cpus in (1, 2, 4, 7, 16, 64):                        # odd log_cpus are rounded down to even (:8-12)
        r = a.copy(); O.best_fft_radix_4(r, w, k, cpus)
        assert np.array_equal(r, ref), cpus
    with pytest.raises(ValueError):
        O.parallel_fft_radix_4(a.copy(), w, k, 1)            # assert!(log_cpus % 2 == 0)
    with pytest.raises(ValueError):
        O.parallel_fft_radix_4(a.copy(), w, k, log_n + 2)    # assert!(log_n >= log_cpus)
    if log_n <= 8:
        canon = canon_list(F, a)
        wc = F.from_mont(w)
        assert canon_list(F, ref) == P.serial_fft_radix_4(F, canon, wc) == P.parallel_fft_radix_4(F, canon, wc, 2 if log_n >= 2 else 0)
    for log_f in range(1, log_n + 1):
        factor = 1 << log_f
        z = a.copy(); z[n // factor:] = 0
        full = z.copy(); O.serial_fft(full, w, k)
        for log_cpus in (0, 1, 2, 3):
            if log_cpus <= log_n:
                r = z.copy(); O.parallel_lde(r, w, k, log_cpus, factor)
                assert np.array_equal(r, full), (log_f, log_cpus)
        for cpus in (1, 3, 8, 1 << (log_n + 1)):             # the last one: log_n <= log_cpus -> serial_lde (:8-9)
            r = z.copy(); O.best_lde(r, w, k, factor, cpus)
            assert np.array_equal(r, full), (log_f, cpus)
        if log_n <= 6:
            canon = canon_list(F, z)
            wc = F.from_mont(w)
            exp = canon_list(F, full)
            assert P.serial_lde(F, canon, wc, factor) == exp
            assert P.parallel_lde(F, canon, wc, min(2, log_n), factor) == exp
            assert P.best_lde(F, canon, wc, factor, 4) == exp
    with pytest.raises(ValueError):
        O.parallel_lde(a.copy(), w, k, log_n + 1, 2)


def test_parallel_radix4_fft_at_the_reference_size(oracles):
    """test_parallel_radix4_fft (src/fft/mod.rs:128-184) at ITS size and over ITS field: 2^22 points of
    experiments::Fr, parallel_fft == parallel_fft_radix_4 == parallel_DIT_fft element for element with the host's own
    worker counts, and the result is the one whose digest tests/golden/fullsize_digests.json commits (the GPU is held to
    the same digest, tests/test_gpu_fullsize.py)."""
    import hashlib
    O = oracles["experiments"]
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fullsize_digests.json")))["reference_tests"]["parallel_radix4_fft"]
    log_n = fx["log_n"]
    n = 1 << log_n
    a = O.gen_elements(0, n, fx["seed"])
    dig = lambda x: hashlib.blake2s(memoryview(np.ascontiguousarray(x)).cast("B"), digest_size=32).hexdigest()
    assert dig(a) == fx["input"]
    _, k, w = O.domain(n)
    log_cpus = O.cpus.bit_length() - 1
    r4 = log_cpus - (log_cpus & 1)
    b, c = a.copy(), a.copy()
    O.parallel_fft(a, w, k, log_cpus)
    O.parallel_fft_radix_4(b, w, k, r4)
    assert np.array_equal(a, b)
    del b
    O.parallel_dit_fft(c, w, k, log_cpus, n)
    assert np.array_equal(a, c)
    assert dig(a) == fx["fft"]


def test_various_ldes_identity(oracles):
    """test_various_ldes (src/polynomials/mod.rs:1084-1130): lde_using_multiple_cosets == filtering_lde (zero-pad +
    best_lde) == fft of the zero-padded vector.  The reference runs it at 2^22 x 16 (three vectors of 2 GiB, about a
    minute of CPU on 8 cores): here at 2^16 x 16 by default and at the reference's size — against the committed digest —
    when HODOR_CPU_FULLSIZE=1 (tests/golden/gen_fullsize.py --ref-sizes asserted the full-size identity when it wrote
    the digest; tests/test_gpu_fullsize.py holds the GPU to it)."""
    import hashlib
    O = oracles["experiments"]
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fullsize_digests.json")))["reference_tests"]["various_ldes"]
    full = os.environ.get("HODOR_CPU_FULLSIZE", "0") not in ("", "0")
    log_n, factor = (fx["log_n"] if full else 16), fx["factor"]
    n = 1 << log_n
    coeffs = O.gen_elements(0, n, fx["seed"])
    coset = O.poly_lde(coeffs, factor)
    _, K, W = O.domain(n * factor)
    filt = np.zeros((n * factor, 4), dtype=np.uint64); filt[:n] = coeffs
    O.best_lde(filt, W, K, factor)
    assert np.array_equal(filt, coset)
    del coset
    naive = np.zeros((n * factor, 4), dtype=np.uint64); naive[:n] = coeffs
    O.best_fft(naive, W, K)
    assert np.array_equal(filt, naive)
    if full:
        assert hashlib.blake2s(memoryview(naive).cast("B"), digest_size=32).hexdigest() == fx["lde"]


def test_fri_by_values_equals_through_coefficients(oracles, field_name):
    """test_one_fri_step / test_fri_on_values_vs_on_coefficients (src/fri/mod.rs:252-361, :510-692):
    every intermediate vector equals the LDE of the folded coefficients a_2i + beta*a_2i+1."""
    O, F = oracles[field_name], PYF[field_name]
    deg, f, outd = 64, 4, 2
    coeffs = O.random_elements(deg, 99)
    lde = O.poly_lde(coeffs, f)
    r = O.fri_commit(lde, f, outd)
    c = canon_list(F, coeffs)
    for i, beta in enumerate(r["challenges"]):
        c = P.fri_fold_coeffs(F, c, F.from_mont(beta))
        nxt = O.poly_lde(mont_array(F, c), f)
        assert np.array_equal(nxt, r["inter_values"][i])
    assert canon_list(F, r["final_coeffs"]) == c[:outd]
    assert r["final_root"] == r["roots"][-1]


def test_fri_through_coefficients_prototype_equals_by_values(oracles, field_name):
    """proof_from_lde_through_coefficients (src/fri/mod.rs:156-248) restated in C and in Python: the reference's own
    assertion — every field of the two prototypes equal (:338-343) — on its test shape (4 coefficients 1, 2, 4, 8,
    lde_factor 4, output_at_degree_plus_one 2, :270-285) and on wider ones, both tree formats."""
    O, F = oracles[field_name], PYF[field_name]
    shapes = [(None, 4, 2), (16, 4, 1), (64, 8, 1), (32, 2, 4), (8, 16, 2)]
    for deg, f, outd in shapes:
        coeffs = mont_array(F, [1, 2, 4, 8]) if deg is None else O.random_elements(deg, 31 + deg)
        lde = O.poly_lde(coeffs, f)
        for combiner in (0, 1):
            if combiner == 1 and f * outd < 4:
                continue
            a = O.fri_commit(lde, f, outd, combiner=combiner)
            b = O.fri_commit(lde, f, outd, combiner=combiner, through_coefficients=True)
            assert a["serialized"] == b["serialized"] and a["roots"] == b["roots"]
            assert a["challenges"] == b["challenges"] and a["final_root"] == b["final_root"]
            assert np.array_equal(a["final_coeffs"], b["final_coeffs"])
            assert len(a["inter_values"]) == len(b["inter_values"])
            assert all(np.array_equal(x, y) for x, y in zip(a["inter_values"], b["inter_values"]))
            if len(lde) <= 256:     # the independent Python twin, small cases
                py = P.fri_commit_through_coefficients(F, canon_list(F, lde), f, outd, combiner=combiner)
                assert P.fri_serialize(F, py) == b["serialized"]
                assert [canon_list(F, v) for v in b["inter_values"]] == py["inter_values"]
    # the verdict of the reference's test body (:287-330): one fold of the coefficients IS the final polynomial
    lde = O.poly_lde(mont_array(F, [1, 2, 4, 8]), 4)
    r = O.fri_commit(lde, 4, 2, through_coefficients=True)
    beta = F.from_mont(r["challenges"][0])
    assert canon_list(F, r["final_coeffs"]) == [(1 + beta * 2) % F.p, (4 + beta * 8) % F.p]
    with pytest.raises(ValueError):
        O.fri_commit(lde, 4, 4, through_coefficients=True)      # no folding step: roots.pop() panics (:226)


def test_value_form_ops_against_bigint(oracles, field_name):
    """test_batch_inversion (src/polynomials/mod.rs:959-985): batch inverse == per-element inverse;
    pointwise ops against Python big-int arithmetic."""
    O, F = oracles[field_name], PYF[field_name]
    n = 257
    a, b = O.random_elements(n, 1), O.random_elements(n, 2)
    ca, cb = canon_list(F, a), canon_list(F, b)
    for op, fn in (("add", lambda x, y: (x + y) % F.p), ("sub", lambda x, y: (x - y) % F.p),
                   ("mul", lambda x, y: x * y % F.p)):
        r = a.copy()
        O.poly_binary(r, b, op)
        assert canon_list(F, r) == [fn(x, y) for x, y in zip(ca, cb)], op
    s = F.to_mont(12345)
    r = a.copy(); O.poly_add_scaled(r, b, s)
    assert canon_list(F, r) == [(x + 12345 * y) % F.p for x, y in zip(ca, cb)]
    for op, fn in (("negate", lambda x: -x % F.p), ("square", lambda x: x * x % F.p),
                   ("pow", lambda x: pow(x, 5, F.p)), ("scale", lambda x: x * 12345 % F.p),
                   ("add_constant", lambda x: (x + 12345) % F.p), ("sub_constant", lambda x: (x - 12345) % F.p)):
        r = a.copy(); O.poly_unary(r, op, c=s, e=5)
        assert canon_list(F, r) == [fn(x) for x in ca], op
    r = a.copy(); O.poly_batch_inversion(r)
    assert canon_list(F, r) == [pow(x, -1, F.p) for x in ca]
    z = a.copy(); z[100] = 0
    before = z.copy()
    with pytest.raises(ValueError):
        O.poly_batch_inversion(z)
    assert np.array_equal(z, before)


def test_degree_one_on_domain_against_bigint(oracles, field_name):
    """evaluate_at_domain_for_degree_one / coset_evaluate_at_domain_for_degree_one (src/polynomials/mod.rs:229-290):
    q(x) = c + alpha x on the domain points w^i (coset: g w^i), against Python big-int arithmetic; equal to
    evaluating the coefficient pair by the oracle's own (coset) FFT; a size that is no power of two is an error."""
    O, F = oracles[field_name], PYF[field_name]
    n = 64
    w = F.domain_generator(n)[0]
    alpha, c = 0x1234567890ABCDEF1234567 % F.p, (F.p - 5)
    for coset in (False, True):
        got = canon_list(F, O.poly_degree_one_on_domain(n, F.to_mont(alpha), F.to_mont(c), coset=coset))
        shift = F.g if coset else 1
        assert got == [(alpha * shift * pow(w, i, F.p) + c) % F.p for i in range(n)]
        q = mont_array(F, [c, alpha] + [0] * (n - 2))
        (O.poly_coset_fft if coset else O.poly_fft)(q)
        assert canon_list(F, q) == got
    with pytest.raises(ValueError):
        O.poly_degree_one_on_domain(48, F.to_mont(alpha), F.to_mont(c))


def test_restated_verifier_accepts_oracle_proofs(oracles):
    """verify_proof_queries (src/fri/verifier.rs:131-289) over a proof assembled from the oracle's commit
    + get_path, the way produce_proof does (src/fri/query_producer.rs:10-53)."""
    O, F = oracles["bn256"], PYF["bn256"]
    coeffs = O.random_elements(32, 3)
    f = 8
    lde = O.poly_lde(coeffs, f)
    n = len(lde)
    r = O.fri_commit(lde, f, 1)
    vectors = [lde] + r["inter_values"]
    for index in (1, 77, n - 1):
        queries, size, idx = [], n, index
        for vec in vectors:
            nodes = O.iop_create(vec)
            ints = array_to_ints(vec)
            for c in sorted([idx, (idx + size // 2) % size]):
                queries.append((c, ints[c], [bytes(x) for x in O.iop_path(nodes, vec, c)]))
            idx = idx if idx < size // 2 else idx - size // 2
            size //= 2
        proof = dict(queries=queries, roots=r["roots"], final_coeffs=array_to_ints(r["final_coeffs"]),
                     initial_degree_plus_one=n // f, lde_factor=f)
        assert P.fri_verify_proof_queries(F, proof, index, array_to_ints(lde)[index])
        assert not P.fri_verify_proof_queries(F, proof, index, array_to_ints(lde)[index] ^ 1)
    with pytest.raises(ValueError):
        P.fri_verify_proof_queries(F, proof, 2, 0)      # even index: lies in the half-size sub-domain
