"""The bounds-checked device build (hodor_amd/csrc/bounds.cuh, `make -C hodor_amd/csrc bounds`): every global load / store
and every LDS slot of every kernel checked against the extents its launcher declares, launchers' declarations checked
against the library's own allocations.  Here: (1) a cross-section of the path runs on it without a single violation and
with the oracle's results, (2) the check FIRES — every extent declared one element short (HODOR_BOUNDS_SHRINK) turns the
kernels' own last accesses into violations the API reports as HODOR_ERR_DEVICE, naming the kernel, (3) a launcher that is
promised more than an allocation of the library's holds is reported on the host.  The whole GPU suite on this build:
bench/bounds_suite.sh, log in profiles/r06/bounds_suite.txt."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "hodor_amd", "libhodor_gpu_bounds.so")

WORKLOAD = r"""
import ctypes as C, numpy as np, hodor_amd
from hodor_amd.handles import Polynomial, IopTree, FriPrototypeHandle, COEFFICIENTS, VALUES
from oracle import pyref as P
from oracle.oracle import Oracle
L = hodor_amd.lib()
L.hodor_bounds_hits.restype = C.c_ulonglong
O = Oracle(P.BN256.p, P.BN256.g)
try:
    ctx = hodor_amd.Context(device=0)
    for log_n in (3, 10, 13, 19):                                  # one tile, one pass, two passes, three passes
        a = O.gen_elements(0, 1 << log_n, 5)
        got, exp = a.copy(), a.copy()
        ctx.poly_coset_fft(got); O.poly_coset_fft(exp)
        assert np.array_equal(got, exp), log_n
        ctx.poly_icoset_fft(got)
        assert np.array_equal(got, a), log_n
    a = O.gen_elements(0, 1 << 12, 6)
    lde = ctx.poly_lde(a, 8)
    assert np.array_equal(lde, O.poly_lde(a, 8))
    for comb in (0, 1):
        proto = ctx.fri_commit(lde, 8, 1, combiner=comb)
        assert proto.serialized == O.fri_commit(lde, 8, 1, combiner=comb)["serialized"]
        proto.free()
    p = Polynomial.from_coeffs(ctx, a)
    v = p.lde(4, coset=True)
    v.square(); v.add_constant(7 * ctx.one % P.BN256.p); v.batch_inversion()
    t = IopTree.create(v)
    root = t.get_root()
    q = v.clone(); q.icoset_fft()
    z = q.evaluate_at(ctx.generator)
    t.free(); q.free(); v.free(); p.free()
    ctx.synchronize()
    print("DONE hits", int(L.hodor_bounds_hits()))
    ctx.close()
except hodor_amd.HodorError as e:
    print("ERROR", e.code, str(e).replace("\n", " "))
"""


def _run(code, **env):
    from conftest import variant_lib
    if variant_lib("bounds", "libhodor_gpu_bounds.so") is None:
        pytest.skip("libhodor_gpu_bounds.so has not been built (make -C hodor_amd/csrc bounds)")
    if env:      # a run that violates on purpose must not count in the suite's own report (bench/bounds_suite.sh)
        env = dict(env, HODOR_BOUNDS_REPORT="")
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HODOR_LIB=LIB, **env), cwd=ROOT, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    return out.stdout.strip().splitlines()[-1], out.stderr


def test_a_cross_section_of_the_path_runs_without_a_violation():
    line, _ = _run(WORKLOAD)
    assert line == "DONE hits 0", line


def test_extents_declared_one_element_short_are_violations_the_api_reports():
    line, _ = _run(WORKLOAD, HODOR_BOUNDS_SHRINK="1", HODOR_SELFTEST="0")
    assert line.startswith("ERROR 3 "), line                        # HODOR_ERR_DEVICE
    assert "bounds build" in line and "device-side" in line and "k_" in line and "range" in line, line


def test_the_start_up_self_test_runs_on_the_checked_kernels_too():
    """with the extents short, the very first thing a context does — its self-test — already trips the check"""
    line, _ = _run(WORKLOAD, HODOR_BOUNDS_SHRINK="1")
    assert line.startswith("ERROR 3 ") and "self-test" in line, line


HOST_SIDE = r"""
import ctypes as C, hodor_amd
L = hodor_amd.lib()
L.hodor_bounds_hits.restype = C.c_ulonglong
ctx = hodor_amd.Context(device=0)
buf = C.c_void_p()
assert L.hodor_buf_alloc(ctx.h, C.c_size_t(1000 * 32), C.byref(buf)) == 0          # 1000 elements ...
before = int(L.hodor_bounds_hits())
rc = L.hodor_poly_fft_dev(ctx.h, None, buf, buf, C.c_uint32(10))                    # ... transformed as if there were 1024
ctx.synchronize_quiet = True
L.hodor_ctx_synchronize(ctx.h)
print("HITS", int(L.hodor_bounds_hits()) - before)
"""


def test_a_launcher_promised_more_than_the_allocation_holds_is_reported_on_the_host():
    line, err = _run(HOST_SIDE, HODOR_BOUNDS_REPORT="")
    assert line.startswith("HITS ") and int(line.split()[1]) >= 1, line
    assert "declares 32768 bytes" in err and "allocation of 32000 bytes" in err, err[-500:]
