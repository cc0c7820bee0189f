"""The start-up self-test of hodor_ctx_create (hodor_amd/csrc/abi_selftest.hip): it passes on a healthy build, it REFUSES
the context — HODOR_ERR_DEVICE, the failing check named through hodor_last_error(NULL) — when a device-side constant or
one word of a twiddle table is corrupted (HODOR_SELFTEST_CORRUPT, a debugging knob that exists for this test), it can be
switched off, and it costs about a millisecond and a half per context (bench/selftest_cost.py records the number)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import hodor_amd
try:
    ctx = hodor_amd.Context(device=0)
except hodor_amd.HodorError as e:
    print("REFUSED", e.code, str(e))
else:
    import numpy as np
    from oracle import pyref as P
    from oracle.oracle import Oracle
    O = Oracle(P.BN256.p, P.BN256.g)
    trips = ctx.host_round_trips()
    a = O.random_elements(1 << 10, 3)
    got, exp = a.copy(), a.copy()
    ctx.poly_fft(got)
    O.poly_fft(exp)
    print("CREATED", "transform-ok" if np.array_equal(got, exp) else "transform-WRONG", trips)
    ctx.close()
"""


def _run(**env):
    out = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, **env), cwd=ROOT, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    return out.stdout.strip().splitlines()[-1]


def test_a_healthy_build_passes_and_the_counters_start_at_zero():
    line = _run()
    assert line.startswith("CREATED transform-ok"), line
    assert line.split()[-1] == "0"          # the self-test's transfers are not the caller's: the counters start at zero


@pytest.mark.parametrize("knob,names", [
    ("1", ("transform", "ifft(fft(x))")),            # 9 x 29 field parameters: the transform kernel computes garbage
    ("2", ("Merkle tree", "root")),                  # BLAKE2s midstate: the device's hashes are not the host's
    ("3", ("transform", "ifft(fft(x))", "FRI")),     # W9 table constants: wrong table words
    ("4", ("transform", "ifft(fft(x))", "FRI", "challenge", "fold")),   # 8 x 32 field parameters
    ("5", ("transform",)),                           # one word of a cached twiddle table
])
def test_a_corrupted_constant_or_table_word_is_refused(knob, names):
    line = _run(HODOR_SELFTEST_CORRUPT=knob)
    assert line.startswith("REFUSED 3 "), line                       # HODOR_ERR_DEVICE, no context
    assert "start-up self-test failed" in line and any(n in line for n in names), line


def test_the_self_test_can_be_switched_off_and_then_the_corruption_goes_unnoticed_at_creation():
    line = _run(HODOR_SELFTEST="0", HODOR_SELFTEST_CORRUPT="1")      # (the knob is read by the self-test only: nothing is corrupted)
    assert line.startswith("CREATED transform-ok"), line
