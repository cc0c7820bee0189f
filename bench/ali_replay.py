#!/usr/bin/env python3
"""What §8(f).1 (value-form operations on the device) buys: the calculate_g operation sequence
(/root/reference/src/ali/per_register/mod.rs:402-526, tests/ali_replay_ref.py) timed
  (B) device-resident: every operation a `_dev` call, one stream, nothing crosses PCIe;
  (A) transform-only offload: coset_lde / icoset_fft through the slice API (host pointers, PCIe both ways
      per call), value-form operations on the host (C port of the reference's loops, 1 thread);
  (P) the PCIe traffic of (A) alone (the 5 uploads + 5 downloads of its transforms, pageable memory).
Results are checked against each other before anything is printed.
    python bench/ali_replay.py [log_n] [factor]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import hodor_amd  # noqa: E402
from ali_replay_ref import DeviceOps, OffloadOps, calculate_g, make_inputs  # noqa: E402
from oracle import pyref as P  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    factor = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    ctx = hodor_amd.Context(hodor_amd.BN256_FR_MODULUS, hodor_amd.BN256_FR_GENERATOR, device=0)
    O = Oracle(P.BN256.p, P.BN256.g)
    witness, consts = make_inputs(O, log_n, factor)

    def dev(x):
        return torch.from_numpy(x.view(np.int64).copy()).cuda()

    d_w = [dev(w) for w in witness]
    d_c = {k: (dev(v) if isinstance(v, np.ndarray) else v) for k, v in consts.items()}
    ops = DeviceOps(ctx)
    got = calculate_g(ops, d_w, factor, d_c)          # warm-up: twiddle tables
    ctx.synchronize()
    reps = 5
    t = time.perf_counter()
    for _ in range(reps):
        got = calculate_g(ops, d_w, factor, d_c)
    ctx.synchronize()
    dev_ms = (time.perf_counter() - t) / reps * 1e3

    def hcopy(d):
        return {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in d.items()}

    off_ops = OffloadOps(O, ctx)
    t = time.perf_counter()
    off = calculate_g(off_ops, [w.copy() for w in witness], factor, hcopy(consts))
    off_ms = (time.perf_counter() - t) * 1e3
    assert np.array_equal(got.cpu().numpy().view(np.uint64), off), "device-resident != transform-only offload"

    n, big = 1 << log_n, (1 << log_n) * factor
    h_small, h_big = np.zeros((n, 4), np.uint64), np.zeros((big, 4), np.uint64)
    d_small, d_big = dev(h_small), dev(h_big)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):                                  # 5 coset_lde calls: n up, n*f down
        d_small.copy_(torch.from_numpy(h_small.view(np.int64)))
        h_big[:] = d_big.cpu().numpy().view(np.uint64)
    d_big.copy_(torch.from_numpy(h_big.view(np.int64)))  # icoset_fft: n*f up, n*f down
    h_big[:] = d_big.cpu().numpy().view(np.uint64)
    torch.cuda.synchronize()
    pcie_ms = (time.perf_counter() - t) * 1e3

    print("calculate_g replay, witness 2 x 2^%d, constraint domain 2^%d (factor %d), src/bn256.rs field" %
          (log_n, log_n + factor.bit_length() - 1, factor))
    print("  (B) device-resident (_dev ABI, 5 coset_lde + 18 value-form ops + icoset_fft): %9.2f ms" % dev_ms)
    print("  (A) transform-only offload (slice API + host value ops, 1 thread):           %9.2f ms" % off_ms)
    print("  (P) PCIe traffic of (A) alone (pageable):                                     %9.2f ms" % pcie_ms)
    print("  (A)/(B) = %.1fx;  (P)/(B) = %.1fx  -- outputs identical" % (off_ms / dev_ms, pcie_ms / dev_ms))
    ctx.close()


if __name__ == "__main__":
    main()
