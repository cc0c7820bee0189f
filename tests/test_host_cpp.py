"""Builds and runs tests/host_cpp/test_host.cpp: the reference's own inline tests (test_worker_size,
test_lde_correctness, make_small_iop, test_one_fri_step) restated in C++ against the host mirror of the
Rust interface (hodor_amd/csrc/host/hodor.hpp), i.e. host C++ -> C ABI -> HIP kernels."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_cpp", "test_host.cpp")


def _build(tmp_path):
    import hodor_amd
    hodor_amd.build()
    exe = str(tmp_path / "test_host")
    libdir = os.path.join(ROOT, "hodor_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", SRC, "-L" + libdir, "-lhodor_gpu",
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


# ---------------------------------------------------------------- plain C (the boundary is a C ABI)
C_SRC = os.path.join(ROOT, "tests", "host_c", "test_abi.c")


def _build_c(tmp_path):
    import hodor_amd
    hodor_amd.build()
    exe = str(tmp_path / "test_abi_c")
    libdir = os.path.join(ROOT, "hodor_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-O2", C_SRC, "-L" + libdir, "-lhodor_gpu",
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
