"""§8(f).1 in use, second caller: the operation sequence of ALI's calculate_deep
(/root/reference/src/ali/per_register/deep.rs:14-146: evaluate_at -> evaluate_at_domain_for_degree_one ->
batch_inversion -> clone / add_constant / scale / mul_assign / add_assign) replayed device-resident through the
`_dev` ABI and compared, bit for bit, with the CPU oracle running the same sequence — h1, h2 and the values at z
that go into the proof."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("log_n,factor,g_factor", [(5, 2, 2), (10, 4, 8), (15, 8, 16), (17, 8, 8)])
def test_calculate_deep_sequence_device_resident_matches_oracle(gpu_ctxs, oracles, field_name, log_n, factor, g_factor, fused):
    """fused: every quotient term as ONE pass (hodor_poly_quotient_term_dev) instead of the reference's five
    value-form operations — the oracle always runs the five."""
    import torch
    from deep_replay_ref import DeviceOps, FusedDeviceOps, OracleOps, calculate_deep, make_inputs
    if field_name != "bn256" and log_n > 10:
        pytest.skip("large cases on the bn256.rs field only")
    ctx, O = gpu_ctxs[field_name], oracles[field_name]
    f_polys, f_ldes, g_poly, g_lde, scalars = make_inputs(O, log_n, factor, g_factor)
    e_h1, e_h2, e_fz, e_gz = calculate_deep(OracleOps(O), f_polys, f_ldes, g_poly, g_lde, scalars)

    def dev(x):
        return torch.from_numpy(x.view(np.int64).copy()).cuda()

    h1, h2, fz, gz = calculate_deep((FusedDeviceOps if fused else DeviceOps)(O, ctx), [dev(p) for p in f_polys], [dev(p) for p in f_ldes], dev(g_poly),
                                    dev(g_lde), scalars)
    ctx.synchronize()
    assert fz == e_fz and gz == e_gz
    assert np.array_equal(h1.cpu().numpy().view(np.uint64), e_h1)
    assert np.array_equal(h2.cpu().numpy().view(np.uint64), e_h2)


@pytest.mark.parametrize("n", [1, 2, 64, 1 << 12, 1 << 16, 1 << 19])
def test_degree_one_on_domain(gpu_ctxs, oracles, field_name, n):
    """evaluate_at_domain_for_degree_one and its coset form (src/polynomials/mod.rs:229-290): both kernel forms
    (running product below 2^16 points, two-level table from there) against the oracle; a size that is not a
    power of two is refused."""
    import torch
    import hodor_amd
    from oracle.oracle import array_to_ints
    ctx, O = gpu_ctxs[field_name], oracles[field_name]
    alpha, c = array_to_ints(O.gen_elements(0, 2, 77 + n))
    for coset in (False, True):
        out = torch.empty((n, 4), dtype=torch.int64, device="cuda")
        ctx.poly_degree_one_on_domain_dev(out, n, alpha, c, coset=coset)
        ctx.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint64), O.poly_degree_one_on_domain(n, alpha, c, coset=coset))
    if n == 64:
        bad = torch.empty((48, 4), dtype=torch.int64, device="cuda")
        with pytest.raises(hodor_amd.HodorError):
            ctx.poly_degree_one_on_domain_dev(bad, 48, alpha, c)
