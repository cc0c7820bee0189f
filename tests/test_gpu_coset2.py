"""GPU parity of the COSET2 tree format (-m gpu): HODOR_COMBINER_COSET2 — the size-2 coset combiner the reference's
README lists as not done (README.md:46; seam = CosetCombiner, src/iop/mod.rs:22-34, only instance
src/iop/trivial_coset_combiner.rs:17-53) — through the C ABI against the CPU oracle and the committed Python
fixtures: trees, queries, FRI prototypes and proofs byte for byte, at BASELINE's sizes through the oracle's digests.
Nothing here reads /root/reference."""
import hashlib
import json
import os

import numpy as np
import pytest

import hodor_amd
from oracle import pyref as P
from oracle.oracle import array_to_ints, ints_to_array

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "coset2_golden.json")))
FULL = json.load(open(os.path.join(HERE, "golden", "fullsize_digests.json")))
PYF = {"bn256": P.BN256, "experiments": P.EXPERIMENTS, "bn254": P.BN254}
T, C2 = hodor_amd.TRIVIAL, hodor_amd.COSET2


def digest(t):
    a = t.cpu().numpy() if hasattr(t, "cpu") else np.ascontiguousarray(t)
    return hashlib.blake2s(memoryview(a).cast("B"), digest_size=32).hexdigest()


@pytest.mark.parametrize("log_n", [2, 3, 4, 6, 9, 10, 11, 12, 13, 16])
def test_coset2_tree_matches_oracle(gpu_ctxs, oracles, field_name, log_n):
    ctx, O = gpu_ctxs[field_name], oracles[field_name]
    n = 1 << log_n
    vals = O.random_elements(n, 1900 + log_n)
    exp = O.iop_create_coset2(vals)
    got = ctx.iop_create_combined(vals, C2)
    assert got.shape == (n // 2, 32) and np.array_equal(got, exp)
    assert not got[0].any()
    root = bytes(got[1])
    ints = array_to_ints(vals)
    for idx in ([0, 1, n - 1, n // 2] if n > 16 else range(n)):
        k = idx % (n // 2)
        path = ctx.iop_path_combined(got, vals, C2, idx)
        assert np.array_equal(path, O.iop_path_coset2(exp, vals, idx))
        assert O.iop_verify_coset2(root, ints[k], ints[k + n // 2], path, k)
        assert ctx.iop_verify_combined(root, [ints[k], ints[k + n // 2]], path, idx, n, C2)
    # TRIVIAL through the combined entry point is the reference format
    assert np.array_equal(ctx.iop_create_combined(vals, T), O.iop_create(vals))
    for bad in (np.zeros((2, 4), np.uint64), np.zeros((6, 4), np.uint64)):
        with pytest.raises(hodor_amd.HodorError):
            ctx.iop_create_combined(bad, C2)


@pytest.mark.parametrize("log_n", [17, 19, 20, 21, 22])
def test_coset2_tree_large_matches_oracle(gpu_ctxs, oracles, log_n):
    """Every schedule of merkle.hip (latency only; one / two throughput launches, then latency) with the combined
    leaf launch — every node against the CPU oracle."""
    import torch
    from gpu_inputs import random_elements
    ctx, O = gpu_ctxs["bn256"], oracles["bn256"]
    n = 1 << log_n
    d_l = random_elements(torch, n, 177 + log_n)
    d_n = torch.empty((n // 2, 32), dtype=torch.uint8, device="cuda")
    ctx.iop_create_combined_dev(d_l, n, C2, d_n)
    ctx.synchronize()
    exp = O.iop_create_coset2(d_l.cpu().numpy().view(np.uint64))
    assert np.array_equal(d_n.cpu().numpy(), exp)


@pytest.mark.parametrize("log_n,batch", [(3, 5), (10, 3), (14, 4), (20, 2)])
def test_coset2_batched_commit_matches_single_trees(gpu_ctxs, oracles, log_n, batch):
    """hodor_iop_create_batch_combined_dev: `batch` columns committed in one call (all registers' LDEs,
    src/prover/mod.rs:73-80) = the trees of the columns one by one, which the oracle pins."""
    import torch
    from gpu_inputs import random_elements
    ctx, O = gpu_ctxs["bn256"], oracles["bn256"]
    n = 1 << log_n
    d_l = random_elements(torch, n * batch, 277 + log_n)
    d_n = torch.empty((batch * (n // 2), 32), dtype=torch.uint8, device="cuda")
    ctx.iop_create_batch_combined_dev(d_l, n, batch, C2, d_n)
    ctx.synchronize()
    host = d_l.cpu().numpy().view(np.uint64)
    for j in range(batch):
        assert np.array_equal(d_n[j * (n // 2):(j + 1) * (n // 2)].cpu().numpy(), O.iop_create_coset2(host[j * n:(j + 1) * n])), j
    t_n = torch.empty((batch * n, 32), dtype=torch.uint8, device="cuda")
    ctx.iop_create_batch_combined_dev(d_l, n, batch, T, t_n)
    ctx.synchronize()
    assert np.array_equal(t_n[:n].cpu().numpy(), O.iop_create(host[:n]))


def test_coset2_query_dev_matches_oracle(gpu_ctxs, oracles):
    import torch
    ctx, O = gpu_ctxs["bn256"], oracles["bn256"]
    for log_n in (2, 3, 5, 12):
        n = 1 << log_n
        vals = O.random_elements(n, 160 + log_n)
        nodes = O.iop_create_coset2(vals)
        d_l = torch.from_numpy(vals.view(np.int64)).cuda()
        d_n = torch.from_numpy(nodes).cuda()
        ints = array_to_ints(vals)
        for idx in sorted({0, 1, n - 1, n // 2, (n * 3) // 7}):
            k = idx % (n // 2)
            values, path = ctx.iop_query_combined_dev(d_l, d_n, n, C2, idx)
            assert values == [ints[k], ints[k + n // 2]]
            assert np.array_equal(path, O.iop_path_coset2(nodes, vals, idx)) and len(path) == log_n - 1
            assert O.iop_verify_coset2(bytes(nodes[1]), values[0], values[1], path, k)
        tn = torch.from_numpy(O.iop_create(vals)).cuda()
        values, path = ctx.iop_query_combined_dev(d_l, tn, n, T, 1)
        assert values == [ints[1]] and len(path) == log_n


@pytest.mark.parametrize("log_deg,lde_factor,out_deg", [(2, 4, 2), (3, 4, 1), (4, 2, 2), (5, 32, 2), (6, 8, 1), (8, 16, 2),
                                                        (9, 4, 8), (10, 4, 1), (11, 8, 1), (13, 8, 4), (15, 4, 1)])
def test_coset2_fri_commit_matches_oracle(gpu_ctxs, oracles, field_name, log_deg, lde_factor, out_deg):
    """proof_from_lde_by_values (src/fri/fri_on_values.rs:11-159) with every oracle built by the COSET2 combiner:
    prototype equality field by field, every intermediate vector and every tree."""
    ctx, O = gpu_ctxs[field_name], oracles[field_name]
    coeffs = O.random_elements(1 << log_deg, 155 + log_deg)
    lde = O.poly_lde(coeffs, lde_factor)
    exp = O.fri_commit(lde, lde_factor, out_deg, combiner=1)
    got = ctx.fri_commit(lde, lde_factor, out_deg, combiner=C2)
    assert got.combiner == C2 and got.num_steps == exp["num_steps"]
    assert got.roots == exp["roots"]
    assert got.challenges == exp["challenges"]
    assert got.final_root == exp["final_root"]
    assert np.array_equal(got.final_coeffs, exp["final_coeffs"])
    assert got.serialized == exp["serialized"]
    n = len(lde)
    assert np.array_equal(got.tree_nodes(-1, n // 2), O.iop_create_coset2(lde))
    for i in range(got.num_steps):
        assert np.array_equal(got.intermediate_values(i, n >> (i + 1)), exp["inter_values"][i])
        assert np.array_equal(got.tree_nodes(i, n >> (i + 2)), O.iop_create_coset2(exp["inter_values"][i])), i
    # the folds do not depend on the tree format: only the challenges (roots) differ from the TRIVIAL prototype
    triv = ctx.fri_commit(lde, lde_factor, out_deg)
    assert triv.serialized == O.fri_commit(lde, lde_factor, out_deg)["serialized"] != got.serialized
    triv.free()
    got.free()


def test_coset2_needs_two_leaves_in_the_last_tree(gpu_ctxs, oracles):
    ctx, O = gpu_ctxs["bn256"], oracles["bn256"]
    lde = O.poly_lde(O.random_elements(16, 3), 2)
    with pytest.raises(hodor_amd.HodorError) as e:
        ctx.fri_commit(lde, 2, 1, combiner=C2)              # last vector has 2 values = ONE combined leaf
    assert e.value.code == hodor_amd._lib.ERR_SIZE
    ctx.fri_commit(lde, 2, 1).free()                         # fine in the reference's format
    ctx.fri_commit(lde, 2, 2, combiner=C2).free()


def test_coset2_gpu_directly_against_python_fixtures(gpu_ctxs, field_name):
    """No C oracle in between: trees, FRI prototype bytes and PROOF bytes of the device path equal the committed
    Python big-int + hashlib fixtures (tests/golden/gen_coset2.py)."""
    import torch
    F, ctx = PYF[field_name], gpu_ctxs[field_name]
    seen = set()
    for key, c in GOLD[field_name]["cases"].items():
        if key.startswith("merkle_"):
            vals = ints_to_array([int(v, 16) for v in c["values_mont"]])
            nodes = ctx.iop_create_combined(vals, C2)
            assert [bytes(x).hex() for x in nodes] == c["nodes"]
            seen.add("merkle")
        elif key.startswith("fri_"):
            coeffs = ints_to_array([F.to_mont(int(v, 16)) for v in c["coeffs"]])
            lde = ctx.poly_lde(coeffs, c["factor"])
            d_lde = torch.from_numpy(lde.view(np.int64)).cuda()
            proto = ctx.fri_commit_dev(d_lde, len(lde), c["factor"], c["out_deg"], combiner=C2)
            assert proto.serialized.hex() == c["serialized"]
            proof = proto.produce_proof(d_lde, c["index"])
            assert proof["raw"].hex() == c["proof"]
            value = int(c["expected_value_mont"], 16)
            if c["out_deg"] == 1:
                assert ctx.fri_verify_proof_combined(proof["raw"], C2, c["index"], value) is True
                assert ctx.fri_verify_proof_strict(proof["raw"], len(lde), c["factor"], 1, c["index"], value,
                                                   combiner=C2) is True
                assert P.fri_verify_proof_queries_coset2(F, proof, c["index"], value) is True
            proto.free()
            seen.add("fri")
    assert seen == {"merkle", "fri"}


@pytest.mark.parametrize("log_deg,lde_factor,out_deg,index", [(3, 4, 1, 5), (8, 8, 2, 777), (12, 8, 1, 31001), (14, 4, 1, 65535)])
def test_coset2_produce_proof(gpu_ctxs, oracles, log_deg, lde_factor, out_deg, index):
    """produce_proof on a device-resident COSET2 prototype: ONE query per round with both values of the coset and the
    combined leaf's path; bytes equal to the Python restatement's; half the paths of the TRIVIAL proof."""
    import torch
    ctx, O, F = gpu_ctxs["bn256"], oracles["bn256"], P.BN256
    coeffs = O.random_elements(1 << log_deg, 19 + log_deg)
    lde = O.poly_lde(coeffs, lde_factor)
    n = len(lde)
    d_lde = torch.from_numpy(lde.view(np.int64)).cuda()
    proto = ctx.fri_commit_dev(d_lde, n, lde_factor, out_deg, combiner=C2)
    ref = O.fri_commit(lde, lde_factor, out_deg, combiner=1)
    proof = proto.produce_proof(d_lde, index)
    assert proof["roots"] == ref["roots"] and len(proof["queries"]) == ref["num_steps"] + 1
    vectors = [lde] + ref["inter_values"]
    size, idx = n, index
    for r, vec in enumerate(vectors):
        lo, hi = sorted([idx, (idx + size // 2) % size])
        ints = array_to_ints(vec)
        qi, qv, qp = proof["queries"][r]
        assert qi == lo and qv == (ints[lo], ints[hi])
        tree = O.iop_create_coset2(vec)
        assert [bytes(x) for x in O.iop_path_coset2(tree, vec, lo)] == qp and len(qp) == size.bit_length() - 2
        idx = idx if idx < size // 2 else idx - size // 2
        size //= 2
    value = array_to_ints(lde[index:index + 1])[0]
    if out_deg == 1:
        assert ctx.fri_verify_proof_combined(proof["raw"], C2, index, value) is True
        assert ctx.fri_verify_proof_combined(proof["raw"], C2, index, value ^ 1) is False
        assert ctx.fri_verify_proof_strict(proof["raw"], n, lde_factor, 1, index, value, combiner=C2) is True
        assert P.fri_verify_proof_queries_coset2(F, proof, index, value) is True
    if out_deg == 1:
        assert proto.verify_prototype(d_lde, index) is True
    triv = ctx.fri_commit_dev(d_lde, n, lde_factor, out_deg)
    t_raw = triv.produce_proof(d_lde, index)["raw"]
    assert len(proof["raw"]) < 0.62 * len(t_raw)
    triv.free()
    proto.free()


@pytest.mark.parametrize("log_n", sorted(int(k) for k, v in FULL["lde"].items() if "coset2_root" in v))
def test_coset2_config2_commit_every_node(gpu_ctxs, log_n):
    """BASELINE config[2]'s codeword (lde(8) of 2^22 coefficients) committed in the COSET2 format: root and the digest
    of all 2^24 nodes against the CPU oracle's (tests/golden/gen_fullsize.py --coset2)."""
    import torch
    ctx, e = gpu_ctxs["bn256"], FULL["lde"][str(log_n)]
    n, f = 1 << log_n, e["factor"]
    a = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    ctx.gen_elements_dev(a, 0, n, e["seed"])
    out = torch.empty((n * f, 4), dtype=torch.int64, device="cuda")
    nodes = torch.empty((n * f // 2, 32), dtype=torch.uint8, device="cuda")
    ctx.poly_lde_dev(a, out, log_n, f)
    ctx.iop_create_combined_dev(out, n * f, C2, nodes)
    ctx.synchronize()
    assert digest(out) == e["lde"]
    assert bytes(nodes[1].cpu().numpy()).hex() == e["coset2_root"]
    assert digest(nodes) == e["coset2_nodes"]
    del a, out, nodes
    torch.cuda.empty_cache()


@pytest.mark.parametrize("log_n", sorted(int(k) for k in FULL.get("fri_coset2", {})))
def test_coset2_config3_fri_commit_bytes(gpu_ctxs, log_n):
    """BASELINE config[3]'s 2^26 codeword through the COSET2 commit: prototype bytes identical to the CPU oracle's."""
    import torch
    ctx, e = gpu_ctxs["bn256"], FULL["fri_coset2"][str(log_n)]
    f = e["factor"]
    log_deg = log_n - (f.bit_length() - 1)
    a = torch.empty((1 << log_deg, 4), dtype=torch.int64, device="cuda")
    ctx.gen_elements_dev(a, 0, 1 << log_deg, e["seed"])
    code = torch.empty((1 << log_n, 4), dtype=torch.int64, device="cuda")
    ctx.poly_lde_dev(a, code, log_deg, f)
    ctx.synchronize()
    assert digest(code) == e["codeword"]
    proto = ctx.fri_commit_dev(code, 1 << log_n, f, e["out_deg_plus_one"], combiner=C2)
    assert proto.num_steps == e["num_steps"]
    assert proto.serialized.hex() == e["serialized"]
    assert proto.final_root.hex() == e["final_root"]
    n = 1 << log_n
    for index in (1, n // 2 + 12345, n - 1):
        proof = proto.produce_proof(code, index)
        value = array_to_ints(code[index:index + 1].cpu().numpy().view(np.uint64))[0]
        assert ctx.fri_verify_proof_strict(proof["raw"], n, f, 1, index, value, combiner=C2) is True
        assert ctx.fri_verify_proof_combined(proof["raw"], C2, index, value ^ 1) is False
    proto.free()
    del a, code
    torch.cuda.empty_cache()


@pytest.mark.parametrize("log_code", [6, 10, 13, 17])
def test_coset2_repeated_commits_are_identical(gpu_ctxs, log_code):
    import torch
    from gpu_inputs import random_elements
    ctx = gpu_ctxs["bn256"]
    f, log_deg = 8, log_code - 3
    n = 1 << log_code
    d_c = random_elements(torch, 1 << log_deg, 131 + log_code)
    d_lde = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    ctx.poly_lde_dev(d_c, d_lde, log_deg, f)
    seen = set()
    for _ in range(25):
        p = ctx.fri_commit_dev(d_lde, n, f, 1, combiner=C2)
        step = p.num_steps // 2
        seen.add((p.serialized, p.intermediate_values(step, n >> (step + 1)).tobytes(),
                  p.tree_nodes(step, n >> (step + 2)).tobytes()))
        p.free()
    assert len(seen) == 1


# ---------------------------------------------------------------- COSET2 trees across ranks (hodor_amd/distributed.py)
@pytest.mark.parametrize("coset", [False, True])
def test_coset2_commit_by_cosets_single_rank_on_device(gpu_ctxs, oracles, coset):
    """lde_commit_by_cosets_distributed(combiner=COSET2) with the HIP backends at world = 1: the values are the fused LDE's
    (a paired block of ONE rank is the natural order) and the tree is the single-device COSET2 tree."""
    import torch
    from hodor_amd.distributed import HipTreeBackend, lde_commit_by_cosets_distributed
    from hodor_amd.sixstep import HipBackend
    ctx, O = gpu_ctxs["bn256"], oracles["bn256"]
    log_n, factor = 14, 8
    n = 1 << log_n
    coeffs = O.random_elements(n, 31337)
    d = torch.from_numpy(coeffs.view(np.int64)).cuda()
    _, _, Omega = O.domain(n * factor)
    shift = O.const("generator") if coset else None
    lde, root, nodes, top = lde_commit_by_cosets_distributed(HipBackend(ctx), HipTreeBackend(ctx), d, log_n, factor,
                                                             Omega, 0, 1, coset_shift=shift, combiner=1)
    ctx.synchronize()
    exp = O.poly_lde(coeffs, factor, coset)
    assert np.array_equal(lde.cpu().numpy().view(np.uint64), exp)
    exp_nodes = O.iop_create_coset2(exp)
    assert root == bytes(exp_nodes[1]) and np.array_equal(nodes.cpu().numpy()[1:], exp_nodes[1:])


@pytest.mark.parametrize("world", [2, 4, 8])
def test_coset2_subtrees_of_paired_blocks_compose_the_global_tree(gpu_ctxs, oracles, world):
    """The device side of the multi-rank COSET2 commit with the ranks played one after the other (the exchange that
    hands out the paired blocks runs for real in tests/test_sixstep_cpu.py): the COSET2 tree that rank d builds over its
    paired block — natural values [d B/2, (d+1) B/2) then N/2 + the same — is subtree d of the single-device COSET2 tree,
    node for node, and the P roots hash up to its root."""
    import torch
    from hodor_amd.distributed import HipTreeBackend, global_node_index
    ctx, O = gpu_ctxs["bn256"], oracles["bn256"]
    big = 1 << 13
    values = O.random_elements(big, 4242)
    nodes = O.iop_create_coset2(values)
    tb = HipTreeBackend(ctx)
    hb = big // world // 2
    roots = []
    for r in range(world):
        block = np.concatenate([values[r * hb:(r + 1) * hb], values[big // 2 + r * hb:big // 2 + (r + 1) * hb]])
        ln = tb.tree(torch.from_numpy(block.view(np.int64)).cuda(), 1)
        ctx.synchronize()
        ln = ln.cpu().numpy()
        assert ln.shape[0] == hb
        w = hb // 2
        while w >= 1:
            idx = [global_node_index(w + j, w, r, world) for j in range(w)]
            assert np.array_equal(ln[w:2 * w], nodes[idx]), (r, w)
            w //= 2
        roots.append(bytes(ln[1]))
    level = roots
    while len(level) > 1:
        level = [ctx.hash_node(level[2 * i], level[2 * i + 1]) for i in range(len(level) // 2)]
    assert level[0] == bytes(nodes[1])
