"""The planner under every shape its knobs can produce: radix caps 2^2 .. 2^11, tiles of 2^6 .. 2^11 elements,
0 .. 16 tile columns, both LDS twiddle-table forms and fixed workgroup sizes — transforms, in-place transforms,
(coset) LDEs and the 4-step building blocks (several passes down the columns and along the rows, chunked, 1-8
ranks) of 2^1 .. 2^17 points must stay bit-identical to the CPU oracle whatever the plan (the knobs
are tuning aids read once per process, so each setting runs in its own process: tests/plan_fuzz_worker.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))

SETTINGS = [
    ({"HODOR_MAX_LOG_R": "2", "HODOR_TILE_LOG": "6"}, "1,2,3,7,8,11", "8:2:1,11:4:0"),
    ({"HODOR_MAX_LOG_R": "3", "HODOR_TILE_LOG": "6", "HODOR_MIN_LOG_C": "0"}, "4,6,9,10,13", "10:1:2,13:2:2,13:8:1"),
    ({"HODOR_MAX_LOG_R": "5", "HODOR_TILE_LOG": "7", "HODOR_MIN_LOG_C": "4"}, "5,8,12,15", "12:4:1,15:2:3"),
    ({"HODOR_MAX_LOG_R": "6", "HODOR_TILE_LOG": "11", "HODOR_NTT_TW_SUB": "0"}, "6,11,13,17"),
    ({"HODOR_MAX_LOG_R": "7", "HODOR_TILE_LOG": "9", "HODOR_MIN_LOG_C": "3"}, "7,9,14,16", "16:4:2"),
    ({"HODOR_MAX_LOG_R": "8", "HODOR_TILE_LOG": "8", "HODOR_MIN_LOG_C": "1", "HODOR_NTT_THREADS": "64"}, "8,10,16"),
    ({"HODOR_MAX_LOG_R": "9", "HODOR_TILE_LOG": "11", "HODOR_NTT_THREADS": "512"}, "9,12,17"),
    ({"HODOR_MAX_LOG_R": "10", "HODOR_TILE_LOG": "10", "HODOR_MIN_LOG_C": "0", "HODOR_TW_HI_MAX_LOG": "0"}, "10,13,15"),
    ({"HODOR_MAX_LOG_R": "11", "HODOR_TILE_LOG": "11", "HODOR_TW_HI_MAX_LOG": "20"}, "11,12,14,17"),
    ({"HODOR_MAX_LOG_R": "11", "HODOR_TILE_LOG": "11", "HODOR_MIN_LOG_C": "2", "HODOR_NTT_THREADS": "128"}, "11,12,16"),
    # the wave-uniform W9 steps off / on without the skipped products, the generic Montgomery digit (HODOR_NTT_P1 = 0:
    # the v_mul_lo path every modulus that is not 1 mod 2^29 takes), and a workgroup size that is not a multiple of 64
    # (rounded down to whole waves by the launcher: the W9 steps deal their work per wave)
    ({"HODOR_NTT_W9": "0"}, "8,9,16,17", "16:2:1"),
    ({"HODOR_NTT_W9": "1", "HODOR_NTT_P1": "0"}, "8,9,16,17", "16:4:1"),
    ({"HODOR_NTT_THREADS": "96", "HODOR_MAX_LOG_R": "8"}, "8,12,16"),
    ({"HODOR_NTT_THREADS": "200", "HODOR_NTT_P1": "0"}, "9,13,17"),
    # W9 steps off: every step after the first is a W3 step and takes the lazy (un-carried) sums of the one before it,
    # three or four hand-overs per pass at radix 2^10 / 2^11
    ({"HODOR_NTT_W9": "0", "HODOR_MAX_LOG_R": "10"}, "10,12,17"),
    ({"HODOR_NTT_W9": "0", "HODOR_MAX_LOG_R": "11", "HODOR_TILE_LOG": "11"}, "11,15,17"),
]


@pytest.mark.parametrize("idx", range(len(SETTINGS)))
def test_transforms_are_plan_independent(idx):
    knobs, logs = SETTINGS[idx][:2]
    six = SETTINGS[idx][2] if len(SETTINGS[idx]) > 2 else ""      # 4-step cases "log_n:world:log_chunks" (multi-pass column mode)
    env = dict(os.environ)
    env.update(knobs)
    out = subprocess.run([sys.executable, os.path.join(HERE, "plan_fuzz_worker.py"), logs, six], capture_output=True,
                         text=True, timeout=900, env=env)
    assert out.returncode == 0 and "PLAN-FUZZ-OK" in out.stdout, (knobs, out.stdout[-1500:], out.stderr[-3000:])


COMMIT_SETTINGS = [
    ({"HODOR_MERKLE_TAIL_LOG": "0", "HODOR_MERKLE_LAT_LOG": "0", "HODOR_FRI_TAIL": "0", "HODOR_FRI_FUSE_FOLD": "0",
      "HODOR_BATCHINV_SEQ": "2"}, "1,2,5,9,12,16"),
    ({"HODOR_MERKLE_TAIL_LOG": "3", "HODOR_MERKLE_LAT_LOG": "8", "HODOR_FRI_FUSE_FOLD": "2", "HODOR_BATCHINV_SEQ": "3"},
     "3,6,8,11,13,17"),
    ({"HODOR_MERKLE_TAIL_LOG": "9", "HODOR_MERKLE_LAT_LOG": "30", "HODOR_FRI_FUSE_FOLD": "1", "HODOR_BATCHINV_SEQ": "64"},
     "4,7,10,12,15"),
    ({"HODOR_MERKLE_TAIL_LOG": "6", "HODOR_MERKLE_LAT_LOG": "11", "HODOR_FRI_TAIL": "0", "HODOR_FRI_FUSE_FOLD": "1",
      "HODOR_BATCHINV_SEQ": "17"}, "5,10,13,14,18"),
    ({"HODOR_MERKLE_TAIL_LOG": "30", "HODOR_MERKLE_LAT_LOG": "14", "HODOR_FRI_FUSE_FOLD": "2"}, "6,9,11,16"),
]


@pytest.mark.parametrize("idx", range(len(COMMIT_SETTINGS)))
def test_trees_fri_commits_and_inversions_are_schedule_independent(idx):
    """Merkle throughput / latency schedules and their hand-over levels, the fused FRI tail, the fold inside the
    leaf launch, the batch-inversion fan-in: whatever the knobs select, the bytes are the oracle's."""
    knobs, logs = COMMIT_SETTINGS[idx]
    env = dict(os.environ)
    env.update(knobs)
    out = subprocess.run([sys.executable, os.path.join(HERE, "commit_fuzz_worker.py"), logs], capture_output=True,
                         text=True, timeout=900, env=env)
    assert out.returncode == 0 and "COMMIT-FUZZ-OK" in out.stdout, (knobs, out.stdout[-1500:], out.stderr[-3000:])


def test_pass_kernel_without_late_kernel_arguments():
    """`make nolate`: k_ntt_pass with its store phase reading the argument struct the ordinary way instead of re-reading it
    from the kernel-argument segment (ntt.hip: HODOR_NO_LATE_ARGS).  The default build rests on the struct sitting at
    offset 0 of that segment; this build does not, and must give the same bytes — a compiler or code-object change that
    broke the assumption would show up as a difference between the two (round-4 advisor finding)."""
    root = os.path.dirname(HERE)
    lib = os.path.join(root, "hodor_amd", "libhodor_gpu_nolate.so")
    from conftest import variant_lib
    assert variant_lib("nolate", "libhodor_gpu_nolate.so") == lib, "libhodor_gpu_nolate.so is stale and cannot be rebuilt here"
    env = dict(os.environ, HODOR_LIB=lib)
    out = subprocess.run([sys.executable, os.path.join(HERE, "plan_fuzz_worker.py"), "1,5,9,10,13,16,17,20", "16:4:1,13:2:2"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "PLAN-FUZZ-OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
