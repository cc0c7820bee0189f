"""COSET2 tree format (include/hodor_gpu.h, HODOR_COMBINER_COSET2 — the size-2 coset combiner the reference's README
lists as not done, README.md:46; seam = CosetCombiner, src/iop/mod.rs:22-34) on the CPU: the C oracle against the
committed Python fixtures, the library's host-side entry points (path, verify, both FRI verifiers: no device needed)
against both, and the properties that make it a valid instance of the trait.  Nothing here needs a GPU."""
import json
import os

import numpy as np
import pytest

import hodor_amd
from hodor_amd import _lib
from oracle import pyref as P
from oracle.oracle import array_to_ints, ints_to_array

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "coset2_golden.json")))
PYF = {"bn256": P.BN256, "experiments": P.EXPERIMENTS, "bn254": P.BN254}
T, C2 = hodor_amd.TRIVIAL, hodor_amd.COSET2


def test_index_maps_are_a_bijection_that_makes_cosets_neighbours():
    """CosetCombiner::{natural_index_into_tree_index, tree_index_into_natural_index, get_coset_for_natural_index}
    (src/iop/mod.rs:22-34): the two members of {i, i + n/2} (trivial_coset_combiner.rs:29-35) sit in ONE leaf."""
    for n in (4, 8, 64):
        seen = set()
        for i in range(n):
            t = P.coset2_natural_to_tree(i, n)
            assert P.coset2_tree_to_natural(t, n) == i
            seen.add(t)
            pair = (i + n // 2) % n
            assert P.coset2_natural_to_tree(pair, n) == t ^ 1          # same leaf, other half
            assert t >> 1 == min(i, pair)                               # leaf index = the smaller member
        assert seen == set(range(n))


def test_oracle_and_host_helpers_match_the_python_fixtures(oracles, field_name):
    F, O = PYF[field_name], oracles[field_name]
    ctx = hodor_amd.Context(F.p, F.g, device=-1)
    for key, case in GOLD[field_name]["cases"].items():
        if not key.startswith("merkle_"):
            continue
        vals = [int(v, 16) for v in case["values_mont"]]
        n = len(vals)
        arr = ints_to_array(vals)
        nodes = O.iop_create_coset2(arr)
        assert [bytes(x).hex() for x in nodes] == case["nodes"]
        assert nodes.shape == (n // 2, 32) and not nodes[0].any()
        root = bytes(nodes[1])
        for idx, name in ((3 % n, "path_3"), (n - 1, "path_last")):
            exp = case[name]
            assert [bytes(x).hex() for x in O.iop_path_coset2(nodes, arr, idx)] == exp
            got = ctx.iop_path_combined(nodes, arr, C2, idx)
            assert [bytes(x).hex() for x in got] == exp and len(exp) == n.bit_length() - 2
            k = idx % (n // 2)
            pair = [vals[k], vals[k + n // 2]]
            assert ctx.hash_leaf_combined(pair, C2) == O.hash_leaf_pair(*pair) == P.hash_leaf_pair(*pair)
            assert ctx.iop_verify_combined(root, pair, got, idx, n, C2) is True
            assert O.iop_verify_coset2(root, pair[0], pair[1], got, k) is True
            assert ctx.iop_verify_combined(root, pair[::-1], got, idx, n, C2) is False          # halves swapped
            assert ctx.iop_verify_combined(root, [pair[0] ^ 1, pair[1]], got, idx, n, C2) is False
            assert ctx.iop_verify_combined(root, pair, got, idx ^ 1, n, C2) is False             # wrong leaf
        # the TRIVIAL branch of the combined entry points is the reference format, untouched
        tn = O.iop_create(arr)
        assert np.array_equal(ctx.iop_path_combined(tn, arr, T, 1), O.iop_path(tn, arr, 1))
        assert ctx.hash_leaf_combined([vals[0]], T) == O.hash_leaf(vals[0])
    with pytest.raises(_lib.HodorError):
        ctx.iop_path_combined(np.zeros((1, 32), np.uint8), ints_to_array([1, 2]), C2, 0)        # n < 4
    ctx.close()


def test_fri_commit_oracle_matches_fixtures_and_verifiers_agree(oracles, field_name):
    F, O = PYF[field_name], oracles[field_name]
    ctx = hodor_amd.Context(F.p, F.g, device=-1)
    for key, case in GOLD[field_name]["cases"].items():
        if not key.startswith("fri_"):
            continue
        coeffs = [int(v, 16) for v in case["coeffs"]]
        f, od, idx = case["factor"], case["out_deg"], case["index"]
        lde = O.poly_lde(ints_to_array([F.to_mont(c) for c in coeffs]), f)
        n = len(lde)
        ref = O.fri_commit(lde, f, od, combiner=1)
        assert ref["serialized"].hex() == case["serialized"]
        assert ref["serialized"] != O.fri_commit(lde, f, od)["serialized"]                       # a different format
        raw = bytes.fromhex(case["proof"])
        value = int(case["expected_value_mont"], 16)
        assert value == array_to_ints(lde[idx:idx + 1])[0]
        if od == 1:
            assert case["verifies"] is True
            assert ctx.fri_verify_proof_combined(raw, C2, idx, value) is True
            assert ctx.fri_verify_proof_combined(raw, C2, idx, value ^ 1) is False
            assert ctx.fri_verify_proof_strict(raw, n, f, od, idx, value, combiner=C2) is True
            assert ctx.fri_verify_proof_strict(raw, n, f * 2, od, idx, value, combiner=C2) is False
            assert ctx.fri_verify_proof_strict(raw, n * 2, f, od, idx, value, combiner=C2) is False
            with pytest.raises(_lib.HodorError):
                ctx.fri_verify_proof(raw, idx, value)                                            # not a TRIVIAL proof
    ctx.close()


def test_coset2_verifiers_refuse_tampered_and_truncated_proofs():
    """The same forgeries tests/test_abi_cpu.py runs against the TRIVIAL format."""
    F = P.BN256
    ctx = hodor_amd.Context(F.p, F.g, device=-1)
    log_deg, f = 4, 4
    coeffs = [pow(7, 31 + i, F.p) for i in range(1 << log_deg)]
    lde = P.poly_lde(F, coeffs, f)
    n = len(lde)
    proto = P.fri_commit(F, lde, f, 1, combiner=P.COSET2)
    for index in (1, n // 2 + 5, n - 1):
        proof = P.fri_produce_proof(F, proto, lde, index, f, 1, combiner=P.COSET2)
        raw = P.fri_proof_to_bytes(proof)
        expected = F.to_mont(lde[index])
        assert len(proof["queries"]) == len(proof["roots"]) == log_deg + 1
        assert P.fri_verify_proof_queries_coset2(F, proof, index, expected) is True
        assert ctx.fri_verify_proof_combined(raw, C2, index, expected) is True
        assert ctx.fri_verify_proof_strict(raw, n, f, 1, index, expected, combiner=C2) is True

        def variant(mut):
            bad = dict(proof, queries=list(proof["queries"]), roots=list(proof["roots"]),
                       final_coeffs=list(proof["final_coeffs"]))
            mut(bad)
            return bad
        def m_lo(b): q = b["queries"][1]; b["queries"][1] = (q[0], (q[1][0] ^ 2, q[1][1]), q[2])
        def m_hi(b): q = b["queries"][2]; b["queries"][2] = (q[0], (q[1][0], q[1][1] ^ 2), q[2])
        def m_path(b): q = b["queries"][1]; b["queries"][1] = (q[0], q[1], [bytes(32)] + list(q[2][1:]))
        def m_root(b): b["roots"][1] = bytes(32)
        def m_final(b): b["final_coeffs"][0] ^= 4
        for mut in (m_lo, m_hi, m_path, m_root, m_final):
            bad = variant(mut)
            assert P.fri_verify_proof_queries_coset2(F, bad, index, expected) is False
            assert ctx.fri_verify_proof_combined(P.fri_proof_to_bytes(bad), C2, index, expected) is False
        # the index of a query must be the smaller member of the coset ("invalid tree index")
        def m_index(b): q = b["queries"][0]; b["queries"][0] = (q[0] + n // 2, q[1], q[2])
        with pytest.raises(ValueError):
            P.fri_verify_proof_queries_coset2(F, variant(m_index), index, expected)
        with pytest.raises(_lib.HodorError):
            ctx.fri_verify_proof_combined(P.fri_proof_to_bytes(variant(m_index)), C2, index, expected)
        # shape attacks are for the strict verifier: fewer rounds, extra final coefficients, a shortened path
        def drop_last_round(b): b["queries"] = b["queries"][:-1]; b["roots"] = b["roots"][:-1]
        def extra_final(b): b["final_coeffs"] = b["final_coeffs"] + [0]
        def short_path(b): q = b["queries"][0]; b["queries"][0] = (q[0], q[1], list(q[2][:-1]))
        for mut in (drop_last_round, extra_final, short_path):
            assert ctx.fri_verify_proof_strict(P.fri_proof_to_bytes(variant(mut)), n, f, 1, index, expected,
                                               combiner=C2) is False
        for cut in (0, 7, 8, 100, len(raw) - 1):
            with pytest.raises(_lib.HodorError):
                ctx.fri_verify_proof_combined(raw[:cut] if cut else b"\x00", C2, index, expected)
        with pytest.raises(_lib.HodorError):
            ctx.fri_verify_proof_combined(raw + b"\x00", C2, index, expected)
    # proof size: one path per round instead of two
    t_raw = P.fri_proof_to_bytes(P.fri_produce_proof(F, P.fri_commit(F, lde, f, 1), lde, 1, f, 1))
    assert len(raw) < 0.6 * len(t_raw)
    ctx.close()


def test_coset2_opening_binds_the_depth_and_canonical_values():
    """A COSET2 leaf is 64 bytes like the input of a node hash.  Round-4 advisor finding: with ONE hash for both, the two
    child digests of an interior node, presented as a "coset value pair" with a shortened path, hash their way to the
    root.  Two independent protections since round 5: COSET2 leaves hash under a personalisation of their own
    ("Shaftoe2": the forged "leaf" no longer equals the node), and hodor_iop_verify_combined refuses every path whose
    length is not log2(n) - 1 and values that are not canonical residues."""
    F = P.BN256
    ctx = hodor_amd.Context(F.p, F.g, device=-1)
    n = 64
    vals = [F.to_mont(pow(3, i, F.p)) for i in range(n)]
    nodes = P.iop_create_coset2(vals)                     # heap array of the tree over n/2 leaves
    root = nodes[1]
    honest = P.iop_path_coset2(nodes, vals, 5)
    assert ctx.iop_verify_combined(root, [vals[5], vals[5 + n // 2]], honest, 5, n, C2) is True
    # the forgery: the interior node at heap index 8 + j (level of width 8); its children nodes[16 + 2j], nodes[17 + 2j]
    # are presented as the two "values" of a leaf
    j = 3
    left, right = nodes[2 * (8 + j)], nodes[2 * (8 + j) + 1]
    as_values = [int.from_bytes(left, "little"), int.from_bytes(right, "little")]
    short_path, idx = [], 8 + j
    while idx > 1:
        short_path.append(nodes[idx ^ 1])
        idx >>= 1
    assert len(short_path) == len(honest) - 2
    def walk(h):
        k = j
        for sib in short_path:
            h = P.hash_node(h, sib) if k % 2 == 0 else P.hash_node(sib, h)
            k >>= 1
        return h
    assert walk(P.hash_node(left, right)) == root                      # with the NODE hash the walk reaches the root ...
    assert P.hash_leaf_pair(*as_values) != P.hash_node(left, right)    # ... which a COSET2 leaf hash never is
    assert walk(P.hash_leaf_pair(*as_values)) != root
    # and the library refuses the opening on its shape alone: wrong depth (and, for most digests, non-canonical "values")
    assert ctx.iop_verify_combined(root, as_values, short_path, j, n, C2) is False
    assert ctx.iop_verify_combined(root, [vals[5], vals[5 + n // 2]], honest[:-1], 5, n, C2) is False
    assert ctx.iop_verify_combined(root, [vals[5], vals[5 + n // 2]], honest + [bytes(32)], 5, n, C2) is False
    # a non-canonical encoding of a committed value (v + p still fits 256 bits for this field) must not open
    assert vals[5] + F.p < 1 << 256
    assert ctx.iop_verify_combined(root, [vals[5] + F.p, vals[5 + n // 2]], honest, 5, n, C2) is False
    ctx.close()


def test_coset2_has_no_cpu_fallback_and_checks_sizes():
    F = P.BN256
    ctx = hodor_amd.Context(F.p, F.g, device=-1)
    a = ints_to_array([F.to_mont(i + 1) for i in range(8)])
    with pytest.raises(_lib.HodorError) as e:
        ctx.iop_create_combined(a, C2)
    assert e.value.code == _lib.ERR_DEVICE
    with pytest.raises(_lib.HodorError) as e:
        ctx.fri_commit(a, 2, 1, combiner=C2)
    assert e.value.code == _lib.ERR_DEVICE
    with pytest.raises(_lib.HodorError) as e:
        ctx.fri_commit(a, 2, 1, combiner=7)
    assert e.value.code in (_lib.ERR_INVALID, _lib.ERR_DEVICE)
    ctx.close()
