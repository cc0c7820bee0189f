#!/usr/bin/env python3
"""The reference's own benchmark SHAPE through the device-resident path: the cubic-VDF proof run
(/root/reference/src/experiments/cubic_vdf.rs:269-356) is 2^20 rows x 4 registers x LDE 16 and prints per-phase
milliseconds — witness polys -> F LDEs -> F oracles -> G poly -> G LDE -> G oracle -> H1 and H2 (DEEP) -> FRI.  This
tool runs the same phase sequence (tests/prove_shape_ref.py: synthetic trace and constraint system of that shape, the
transcript driving every challenge, query phase included as in Prover::prove) with every polynomial, LDE, tree and
FRI vector resident in HBM, prints the per-phase times in the reference's phase names, the number of host round trips
and the peak HBM, then runs the SAME sequence on the CPU oracle (the C port of the reference's schedules, all host
cores) and requires the two proofs to be byte-identical.
    python bench/prove_shape.py [log_rows=20] [registers=4] [lde_factor=16] [--no-cpu] [--coset2]"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import hodor_amd  # noqa: E402
import prove_shape_ref as ps  # noqa: E402
from oracle import pyref as P  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    log_rows = int(args[0]) if len(args) > 0 else 20
    registers = int(args[1]) if len(args) > 1 else 4
    lde_factor = int(args[2]) if len(args) > 2 else 16
    with_cpu = "--no-cpu" not in sys.argv
    combiner = hodor_amd.COSET2 if "--coset2" in sys.argv else hodor_amd.TRIVIAL   # every oracle in the COSET2 tree format
    ctx = hodor_amd.Context(hodor_amd.BN256_FR_MODULUS, hodor_amd.BN256_FR_GENERATOR, device=0)
    O = Oracle(P.BN256.p, P.BN256.g)
    trace, prep = ps.make_trace(O, log_rows, registers)
    d_trace, d_prep = ps.to_device(trace, prep)

    def clock():
        torch.cuda.synchronize()
        return time.perf_counter()

    dev = ps.DeviceProver(O, ctx, combiner=combiner)
    proof, _, _ = ps.prove(dev, d_trace, d_prep, lde_factor, clock)          # warm-up: twiddle tables, FRI slab
    torch.cuda.reset_peak_memory_stats()
    runs = []
    for _ in range(3):
        dev.host_round_trips = 0
        t = clock()
        again, times, marks = ps.prove(dev, d_trace, d_prep, lde_factor, clock)
        total = clock() - t
        assert again == proof, "the device-resident run is not deterministic"
        runs.append((total, times))
    runs.sort(key=lambda r: r[0])
    total, times = runs[len(runs) // 2]
    peak = torch.cuda.max_memory_allocated() / 2**30
    print("prove-shaped run: %d registers x 2^%d rows, LDE %d (f LDEs 2^%d, g LDE / h2 2^%d points), src/bn256.rs field, %s"
          % (registers, log_rows, lde_factor, log_rows + lde_factor.bit_length() - 1,
             log_rows + 2 + lde_factor.bit_length() - 1,
             "COSET2 oracles (opt-in tree format)" if combiner else "the reference's tree format"))
    print("proof %d bytes, blake2s %s" % (len(proof), hashlib.blake2s(proof, digest_size=32).hexdigest()))
    cpu_times = None
    if with_cpu:
        t = time.perf_counter()
        cpu_proof, cpu_times, cpu_marks = ps.prove(ps.OracleProver(O, P.BN256, combiner=combiner), trace, prep, lde_factor)
        cpu_total = time.perf_counter() - t
        assert cpu_marks == marks, ("phase digests differ", cpu_marks, marks)
        assert cpu_proof == proof, "device-resident proof differs from the CPU port's"
        print("CPU port (oracle/hodor_oracle.c, %d host threads): proof bytes IDENTICAL" % O.cpus)
    print("%-16s %12s %s" % ("phase", "device ms", "CPU port ms" if cpu_times else ""))
    for name in ps.PHASES:
        print("%-16s %12.2f %s" % (name, times[name] * 1e3, ("%12.0f" % (cpu_times[name] * 1e3)) if cpu_times else ""))
    print("%-16s %12.2f %s" % ("total", total * 1e3, ("%12.0f" % (cpu_total * 1e3)) if cpu_times else ""))
    print("host round trips in one run (32-byte roots, evaluations at z, FRI prototypes, query answers): %d; "
          "peak HBM %.1f GiB" % (dev.host_round_trips, peak))
    print("phase digests:", marks)
    ctx.close()


if __name__ == "__main__":
    main()
