#!/usr/bin/env python3
"""ONE of the two `extra` workloads of bench.py, alone and a known number of times, so that a rocprofv3 --pmc pass over this
process can be divided by that number (bench/profile.sh -> profiles/rNN/pmc_traffic.json "extras"):
    python bench/extra_workload.py lde_commit [runs=4]     LDE x8 of 2^22 + Merkle commit   (BASELINE config[2])
    python bench/extra_workload.py fri_commit [runs=4]     FRI commit of the 2^26 codeword  (BASELINE config[3])
Every run is gated on the CPU oracle's committed root / prototype bytes, like bench.py's own legs; the first run also
builds the twiddle tables (kernels the profile summary does not count: it keeps k_ntt_pass, k_merkle_*, k_fri_*)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import hodor_amd  # noqa: E402


def main():
    which = sys.argv[1]
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    ctx = hodor_amd.Context(hodor_amd.BN256_FR_MODULUS, hodor_amd.BN256_FR_GENERATOR, device=0)
    if which == "lde_commit":
        n, big = 1 << bench.LDE_LOG_N, (1 << bench.LDE_LOG_N) * bench.LDE_FACTOR
        fx = bench.FIXTURES["lde"][str(bench.LDE_LOG_N)]
        coeffs = torch.empty((n, 4), dtype=torch.int64, device="cuda")
        ctx.gen_elements_dev(coeffs, 0, n, fx["seed"])
        lde = torch.empty((big, 4), dtype=torch.int64, device="cuda")
        nodes = torch.empty((big, 32), dtype=torch.uint8, device="cuda")
        for _ in range(runs):
            ctx.poly_lde_dev(coeffs, lde, bench.LDE_LOG_N, bench.LDE_FACTOR)
            ctx.iop_create_dev(lde, big, nodes)
            torch.cuda.synchronize()
            assert bytes(nodes[1].cpu().numpy()).hex() == fx["root"]
    elif which == "fri_commit":
        fx = bench.FIXTURES["fri"][str(bench.FRI_LOG_N)]
        factor = fx["factor"]
        log_deg = bench.FRI_LOG_N - (factor.bit_length() - 1)
        n = (1 << log_deg) * factor
        coeffs = torch.empty((1 << log_deg, 4), dtype=torch.int64, device="cuda")
        ctx.gen_elements_dev(coeffs, 0, 1 << log_deg, fx["seed"])
        code = torch.empty((n, 4), dtype=torch.int64, device="cuda")
        ctx.poly_lde_dev(coeffs, code, log_deg, factor)
        torch.cuda.synchronize()
        for _ in range(runs):
            proto = ctx.fri_commit_dev(code, n, factor, 1)
            assert proto.serialized.hex() == fx["serialized"]
            proto.free()
    else:
        raise SystemExit("lde_commit | fri_commit")
    print("%s x %d ok" % (which, runs))
    ctx.close()


if __name__ == "__main__":
    main()
