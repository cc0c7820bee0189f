"""The whole phase sequence of the reference's proof run (cubic_vdf.rs:288-354 / Prover::prove) device-resident
against the same sequence on the CPU oracle: the assembled proof bytes — evaluations at z, all oracle roots, the
oracle queries, both FRI proofs — must be identical (tests/prove_shape_ref.py; bench/prove_shape.py runs the
reference's own shape, 4 registers x 2^20 rows x LDE 16, with per-phase times)."""
import numpy as np
import pytest

from oracle import pyref as P

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_rows,registers,lde_factor", [(6, 2, 4), (10, 4, 16), (12, 3, 8)])
def test_prove_shaped_run_is_byte_identical_to_the_cpu_port(gpu_ctxs, oracles, log_rows, registers, lde_factor):
    import prove_shape_ref as ps
    ctx, O = gpu_ctxs["bn256"], oracles["bn256"]
    trace, prep = ps.make_trace(O, log_rows, registers)
    exp, _, exp_marks = ps.prove(ps.OracleProver(O, P.BN256), [t.copy() for t in trace], ps.copy_prep(prep), lde_factor)
    d_trace, d_prep = ps.to_device(trace, prep)
    dev = ps.DeviceProver(O, ctx)
    got, times, marks = ps.prove(dev, d_trace, d_prep, lde_factor)
    assert marks == exp_marks
    assert got == exp
    assert set(times) == set(ps.PHASES)
    assert dev.host_round_trips > 0
    # ... and the proof VERIFIES: both FRI proofs under the strict verifier (shape bound to the domain, every Merkle path,
    # every fold, the final polynomial) and under the restated reference verifier; every oracle query against its root
    for raw, size, x, value in ps.prove.last["fri"]:
        assert ctx.fri_verify_proof_strict(raw, size, lde_factor, 1, x, value) is True
        assert ctx.fri_verify_proof(raw, x, value) is True
        assert ctx.fri_verify_proof_strict(raw, size, lde_factor, 1, x, value ^ 1) is False
        assert ctx.fri_verify_proof_strict(raw, size, lde_factor // 2, 1, x, value) is False   # the caller's rate binds
    for root, x, (value, path) in ps.prove.last["queries"]:
        assert ctx.iop_verify(root, value, path, x) is True
        assert O.iop_verify(root, value, path, x) is True


@pytest.mark.parametrize("log_rows,registers,lde_factor", [(6, 2, 4), (10, 4, 16)])
def test_prove_shaped_run_with_coset2_oracles(gpu_ctxs, oracles, log_rows, registers, lde_factor):
    """The same phase sequence with every oracle (f, g, both FRI instances) in the COSET2 tree format: device-resident
    proof bytes identical to the CPU oracle's, smaller than the reference-format proof, and every piece verifies —
    both FRI proofs under the strict COSET2 verifier, every oracle query (both values of the coset, one path) against
    its root."""
    import prove_shape_ref as ps
    import hodor_amd
    ctx, O = gpu_ctxs["bn256"], oracles["bn256"]
    trace, prep = ps.make_trace(O, log_rows, registers)
    cp = lambda: ([t.copy() for t in trace], ps.copy_prep(prep))
    exp, _, exp_marks = ps.prove(ps.OracleProver(O, P.BN256, combiner=1), *cp(), lde_factor)
    triv, _, _ = ps.prove(ps.OracleProver(O, P.BN256), *cp(), lde_factor)
    d_trace, d_prep = ps.to_device(trace, prep)
    dev = ps.DeviceProver(O, ctx, combiner=hodor_amd.COSET2)
    got, times, marks = ps.prove(dev, d_trace, d_prep, lde_factor)
    assert marks == exp_marks and got == exp
    assert len(got) < 0.7 * len(triv)
    for raw, size, x, value in ps.prove.last["fri"]:
        assert ctx.fri_verify_proof_strict(raw, size, lde_factor, 1, x, value, combiner=hodor_amd.COSET2) is True
        assert ctx.fri_verify_proof_strict(raw, size, lde_factor, 1, x, value ^ 1, combiner=hodor_amd.COSET2) is False
    for root, x, (values, path) in ps.prove.last["queries"]:
        n = (1 << (len(path) + 1))
        assert ctx.iop_verify_combined(root, list(values), path, x, n, hodor_amd.COSET2) is True
        k = x % (n // 2)
        assert O.iop_verify_coset2(root, values[0], values[1], path, k) is True
