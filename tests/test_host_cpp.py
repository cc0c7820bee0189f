"""Builds and runs tests/host_cpp/test_host.cpp: the reference's own inline tests (test_worker_size,
test_lde_correctness, make_small_iop, test_one_fri_step) restated in C++ against the host mirror of the
Rust interface (hodor_amd/csrc/host/hodor.hpp), i.e. host C++ -> C ABI -> HIP kernels."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_cpp", "test_host.cpp")
# the C / C++ programs link against the library the Python tests load: HODOR_LIB selects a twin build (the bounds-checked
# one, bench/bounds_suite.sh, which sets HODOR_SUITE_LIB) for them too; other twins (asan) are for the Python process only
LIBFLAG = "-l:" + os.path.basename((os.environ.get("HODOR_SUITE_LIB") and os.environ.get("HODOR_LIB")) or "libhodor_gpu.so")


def _build(tmp_path):
    import hodor_amd
    hodor_amd.build()
    exe = str(tmp_path / "test_host")
    libdir = os.path.join(ROOT, "hodor_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", SRC, "-L" + libdir, LIBFLAG,
                           "-Wl,-rpath," + libdir, "-o", exe])
    return exe


def test_host_mirror_compiles_against_the_c_abi(tmp_path):
    """CPU: the C++ host layer needs nothing but include/hodor_gpu.h and the shared library."""
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_reference_tests_in_cpp(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all tests passed" in out.stdout


# ---------------------------------------------------------------- Prover::prove's shape through hodor.hpp only
PS_SRC = os.path.join(ROOT, "tests", "host_cpp", "prove_shape.cpp")


def _build_prove_shape(tmp_path):
    import hodor_amd
    hodor_amd.build()
    exe = str(tmp_path / "prove_shape")
    libdir = os.path.join(ROOT, "hodor_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-pthread", PS_SRC, "-L" + libdir, LIBFLAG,
                           "-Wl,-rpath," + libdir, "-o", exe])
    return exe


def test_prove_shape_replay_compiles_against_the_host_mirror(tmp_path):
    """CPU: the prover-shaped replay uses nothing but hodor.hpp (no `_dev` entry point, device pointer or stream in its
    source) and links against the C ABI alone."""
    assert os.path.exists(_build_prove_shape(tmp_path))
    for path in (PS_SRC, os.path.join(ROOT, "tests", "host_cpp", "ali_instance.hpp")):
        code = "\n".join(l.split("//")[0] for l in open(path).read().splitlines())
        assert "_dev(" not in code and "hipStream" not in code and "dev_ptr" not in code


@pytest.mark.gpu
@pytest.mark.parametrize("log_rows,registers,lde_factor,combiner,ali_mode", [
    (6, 2, 4, 0, 0), (10, 4, 16, 0, 0), (12, 3, 8, 0, 0), (10, 4, 16, 1, 0), (6, 2, 4, 0, 1), (12, 3, 8, 0, 1), (10, 4, 16, 1, 1)])
def test_prove_shape_from_cpp_is_byte_identical_to_the_cpu_port(tmp_path, oracles, log_rows, registers, lde_factor, combiner, ali_mode):
    """The phases of Prover::prove (src/prover/mod.rs:66-174; cubic_vdf.rs:288-354) driven from C++ through the
    device-resident Polynomial / IOP / FRI objects of hodor.hpp — ALIInstance::from_arp's divisor precompute included,
    AS WRITTEN in the reference on Polynomial::as_mut() (ali_mode 0) or device-resident (ali_mode 1): the assembled proof
    bytes equal the bytes the CPU oracle assembles for the same instance from ITS restatement of from_arp
    (tests/prove_shape_ref.py, tests/ali_replay_ref.py), and the library counted no more host round trips than the
    schedule has results to hand over."""
    import json
    import sys

    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import prove_shape_ref as ps
    from oracle import pyref as P
    exe = _build_prove_shape(tmp_path)
    out_bin = str(tmp_path / "proof.bin")
    run = subprocess.run([exe, str(log_rows), str(registers), str(lde_factor), str(combiner), out_bin, "1", "0", str(ali_mode)],
                         capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    line = json.loads(run.stdout.strip().splitlines()[-1])
    O = oracles["bn256"]
    trace, prep = ps.make_trace(O, log_rows, registers)
    exp, _, _ = ps.prove(ps.OracleProver(O, P.BN256, combiner=combiner), trace, prep, lde_factor)
    got = open(out_bin, "rb").read()
    assert len(got) == line["proof_bytes"] == len(exp)
    assert got == exp
    # roots (1 wait for all f oracles + 1 for g), 4 evaluations, 3 batch inversions, 2 prototypes (one wait), 2 FRI proofs,
    # registers + 1 oracle queries
    assert 0 < line["host_round_trips"] <= 2 + 4 + 3 + 1 + 2 + registers + 1      # (both prototypes behind one wait)
    assert set(line["phases_ms"]) == set(ps.PHASES)
    # from_arp: as written the divisor vector (4 n elements) goes up, comes down inverted and goes up again, and the coset
    # table of the adjustment polynomials comes down once; device-resident nothing but the roots and a few inverses move
    big = 32 * 4 * (1 << log_rows)
    fa = line["from_arp"]
    if ali_mode == 0:
        assert fa["h2d_bytes"] >= 2 * big and fa["d2h_bytes"] >= big and line["h2d_bytes"] >= 2 * big   # 2 adjustment polynomials per proof
    else:
        assert fa["h2d_bytes"] < 4096 and fa["d2h_bytes"] < 4096 and line["h2d_bytes"] < (1 << 16)


# ---------------------------------------------------------------- plain C (the boundary is a C ABI)
C_SRC = os.path.join(ROOT, "tests", "host_c", "test_abi.c")


def _build_c(tmp_path):
    import hodor_amd
    hodor_amd.build()
    exe = str(tmp_path / "test_abi_c")
    libdir = os.path.join(ROOT, "hodor_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-O2", C_SRC, "-L" + libdir, LIBFLAG,
                           "-Wl,-rpath," + libdir, "-o", exe])
    return exe


def test_header_is_c11_and_compute_refuses_without_a_device(tmp_path):
    """CPU: include/hodor_gpu.h compiles as strict C11, a C program links against the library, host-side
    helpers work and compute entry points return HODOR_ERR_DEVICE (no CPU fallback)."""
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-fsyntax-only", "-x", "c",
                           os.path.join(ROOT, "include", "hodor_gpu.h")])
    out = subprocess.run([_build_c(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "compute refused as designed" in out.stdout


@pytest.mark.gpu
def test_c_program_on_device(tmp_path):
    out = subprocess.run([_build_c(tmp_path), "0"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all tests passed" in out.stdout
    # a plain C process (no Python, no torch) binds RCCL at run time and runs the 4-step transform through
    # hodor_sixstep_exchange_dev on a one-rank communicator
    assert "through the library's exchange (one-rank RCCL) ok" in out.stdout, out.stdout + out.stderr


# ---------------------------------------------------------------- two plain-C processes over the library's own multi-GPU schedules
D2_SRC = os.path.join(ROOT, "tests", "host_c", "test_dist2.c")


def _build_dist2(tmp_path):
    import hodor_amd
    hodor_amd.build()
    exe = str(tmp_path / "test_dist2")
    libdir = os.path.join(ROOT, "hodor_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-O2", D2_SRC, "-L" + libdir, LIBFLAG,
                           "-Wl,-rpath," + libdir, "-o", exe])
    return exe


def test_two_process_c_program_compiles_as_c11(tmp_path):
    assert os.path.exists(_build_dist2(tmp_path))


@pytest.mark.gpu
def test_two_c_processes_run_the_library_schedules_and_the_transport_soak(tmp_path):
    """fork + hipIpc + the direct transports, no Python, no torch, no RCCL: hodor_dist_ntt_natural_dev,
    hodor_dist_lde_commit_dev (both tree formats) and 10 000 generations of hodor_dist_ntt_forward_dev on alternating
    payloads per transport, every generation compared with its payload's known answer (a stale line anywhere = a
    mismatch).  On this box the two ranks share the one GPU; `test_dist2 G 0 1` is the same program between two devices."""
    out = subprocess.run([_build_dist2(tmp_path), "10000"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "all tests passed" in out.stdout
    assert out.stdout.count("0 mismatches") == 4, out.stdout


# ---------------------------------------------------------------- the host-only half of the mirror, EXECUTED on the CPU
def test_host_only_half_of_the_mirror_runs_without_a_device(tmp_path):
    """tests/host_cpp/test_host_cpu.cpp: Field on a device-less context, Domain, the combiners, FRIProof parse / re-encode,
    NaiveFriIop::verify_proof(_strict), IOP::verify, Transcript — against a fixture written here by the Python restatement
    (oracle/pyref.py) — and the refusal (HODOR_ERR_DEVICE) of the device-side constructors."""
    import hodor_amd
    from oracle import pyref as P
    hodor_amd.build()
    F = P.BN256
    log_deg, factor, index = 3, 4, 7
    coeffs = [pow(11, 20 + i, F.p) for i in range(1 << log_deg)]
    lde = P.poly_lde(F, coeffs, factor)
    proto = P.fri_commit(F, lde, factor, 1)
    raw = P.fri_proof_to_bytes(P.fri_produce_proof(F, proto, lde, index, factor, 1))
    u64 = lambda v: int(v).to_bytes(8, "little")
    t = P.Transcript(F)
    t.commit_bytes(bytes(range(32)))
    t.commit_field_element(12345)
    ch_bytes = t.get_challenge_bytes()
    ch_elem = F.to_mont(t.get_challenge())
    leafs = [F.to_mont(v) for v in lde]
    nodes = P.iop_create(leafs)
    leaf_index = 21
    path = P.iop_path(nodes, leafs, leaf_index)
    fx = (u64(len(raw)) + raw + u64(index) + P.mont_to_bytes(F.to_mont(lde[index])) + u64(len(lde)) + u64(factor)
          + ch_bytes + P.mont_to_bytes(ch_elem) + P.mont_to_bytes(leafs[leaf_index]) + u64(leaf_index) + u64(len(path))
          + b"".join(bytes(x) for x in path) + bytes(nodes[1]))
    fixture = tmp_path / "fixture.bin"
    fixture.write_bytes(fx)
    exe = str(tmp_path / "test_host_cpu")
    libdir = os.path.join(ROOT, "hodor_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-pthread", os.path.join(ROOT, "tests", "host_cpp", "test_host_cpu.cpp"),
                           "-L" + libdir, LIBFLAG, "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe, str(fixture)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "host-only checks passed" in out.stdout, out.stdout + out.stderr
