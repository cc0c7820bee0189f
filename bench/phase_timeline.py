#!/usr/bin/env python3
"""Per-wave phase timeline of k_ntt_pass (ablation build only: HODOR_DBG = 16 | pass << 8 makes wave 0 of
every 64th workgroup of that pass stamp the shader clock at its phase boundaries).  Prints the median
cycles a wave spends in each phase of each of the three passes of a 2^24 transform.
    HODOR_LIB=$PWD/hodor_amd/libhodor_gpu_ablate.so python bench/phase_timeline.py"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = ["stage tw + load + inter-pass tw", "barrier", "step m=1", "barrier", "step m=4", "barrier", "step m=16",
         "barrier", "step m=64", "barrier", "store"]


def one_pass(p):
    import numpy as np
    import torch

    import hodor_amd
    ctx = hodor_amd.Context(hodor_amd.BN256_FR_MODULUS, hodor_amd.BN256_FR_GENERATOR, device=0)
    log_n = 24
    n = 1 << log_n
    a = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    ctx.gen_elements_dev(a, 0, n, 1)
    b = torch.empty_like(a)
    for _ in range(30):
        ctx.poly_fft_dev(a, b, log_n)
    ctx.synchronize()
    lib = C.CDLL(os.environ["HODOR_LIB"])
    st = np.zeros(1024 * 16, dtype=np.uint64)
    rc = lib.hodor_ablate_read_stamps(st.ctypes.data_as(C.c_void_p), C.c_size_t(len(st)))
    assert rc == 0, rc
    st = st.reshape(1024, 16)[:256].astype(np.int64)     # 16384 workgroups / 64
    d = np.diff(st[:, :12], axis=1)
    med = np.median(d, axis=0)
    total = np.median(st[:, 11] - st[:, 0])
    print("pass %d (log_l = %d): wave residency %d cycles (median)" % (p + 1, 8 * p, total))
    for name, v in zip(NAMES, med):
        print("    %-34s %8d  %5.1f %%" % (name, v, 100.0 * v / total))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one_pass(int(sys.argv[1]))
    else:
        for p in range(3):
            env = dict(os.environ, HODOR_DBG=str(16 | (p << 8)))
            subprocess.run([sys.executable, os.path.abspath(__file__), str(p)], env=env, check=True)
