"""§8(f).1 in use: the operation sequence of ALI's calculate_g
(/root/reference/src/ali/per_register/mod.rs:402-526: coset_lde -> pow -> scale/negate -> add_assign ->
add_constant -> mul_assign -> ... -> icoset_fft) replayed device-resident through the `_dev` ABI and
compared, bit for bit, with the CPU oracle running the same sequence; the transform-only-offload form
(slice API + host value ops) must agree too."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_n,factor", [(6, 2), (10, 4), (14, 8)])
def test_calculate_g_sequence_device_resident_matches_oracle(gpu_ctxs, oracles, field_name, log_n, factor):
    import torch
    from ali_replay_ref import DeviceOps, OffloadOps, OracleOps, calculate_g, make_inputs
    if field_name != "bn256" and log_n > 10:
        pytest.skip("large case on the bn256.rs field only")
    ctx, O = gpu_ctxs[field_name], oracles[field_name]
    witness, consts = make_inputs(O, log_n, factor)
    exp = calculate_g(OracleOps(O), [w.copy() for w in witness], factor,
                      {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in consts.items()})

    def dev(x):
        return torch.from_numpy(x.view(np.int64).copy()).cuda()

    d_consts = {k: (dev(v) if isinstance(v, np.ndarray) else v) for k, v in consts.items()}
    got = calculate_g(DeviceOps(ctx), [dev(w) for w in witness], factor, d_consts)
    ctx.synchronize()
    assert np.array_equal(got.cpu().numpy().view(np.uint64), exp)
    if log_n <= 10:
        off = calculate_g(OffloadOps(O, ctx), [w.copy() for w in witness], factor,
                          {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in consts.items()})
        assert np.array_equal(off, exp)
