#!/usr/bin/env python3
"""The DEEP step (calculate_deep, /root/reference/src/ali/per_register/deep.rs:14-146, tests/deep_replay_ref.py)
timed device-resident — three masks over two registers: 4 evaluations at a point, 3 divisor polynomials
(evaluate_at_domain_for_degree_one + batch_inversion), 4 clone / add_constant / mul_assign chains — against the
same sequence on the host by the CPU oracle (C port of the reference's loops, all cores where the reference uses
its Worker), results compared before anything is printed.
    python bench/deep_replay.py [log_n] [factor]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import hodor_amd  # noqa: E402
from deep_replay_ref import DeviceOps, OracleOps, calculate_deep, make_inputs  # noqa: E402
from oracle import pyref as P  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    factor = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    ctx = hodor_amd.Context(hodor_amd.BN256_FR_MODULUS, hodor_amd.BN256_FR_GENERATOR, device=0)
    O = Oracle(P.BN256.p, P.BN256.g)
    f_polys, f_ldes, g_poly, g_lde, scalars = make_inputs(O, log_n, factor, factor)

    def dev(x):
        return torch.from_numpy(x.view(np.int64).copy()).cuda()

    d = ([dev(p) for p in f_polys], [dev(p) for p in f_ldes], dev(g_poly), dev(g_lde))
    ops = DeviceOps(O, ctx)
    got = calculate_deep(ops, *d, scalars)             # warm-up: tables
    ctx.synchronize()
    reps = 5
    t = time.perf_counter()
    for _ in range(reps):
        got = calculate_deep(ops, *d, scalars)
    ctx.synchronize()
    dev_ms = (time.perf_counter() - t) / reps * 1e3
    t = time.perf_counter()
    exp = calculate_deep(OracleOps(O), f_polys, f_ldes, g_poly, g_lde, scalars)
    cpu_ms = (time.perf_counter() - t) * 1e3
    assert got[2] == exp[2] and got[3] == exp[3]
    assert np.array_equal(got[0].cpu().numpy().view(np.uint64), exp[0])
    assert np.array_equal(got[1].cpu().numpy().view(np.uint64), exp[1])
    big = (1 << log_n) * factor
    print("calculate_deep, 2 x 2^%d witness coefficients, LDE domain 2^%d: device-resident %.2f ms; the same sequence "
          "on the host (CPU oracle) %.0f ms; results identical (h1, h2, values at z)"
          % (log_n, big.bit_length() - 1, dev_ms, cpu_ms))


if __name__ == "__main__":
    main()
