import os
import sys

import pytest

# The tests move their inputs and results with torch (`torch.from_numpy(x).cuda()`, `t.cpu()`) from and to PAGEABLE numpy /
# torch memory.  For a large pageable copy the HIP runtime pins the caller's pages itself and keeps the pin in a cache; a later
# small CPU tensor that malloc places inside such a stale range is then copied to DIRECTLY, and when the runtime drops the
# stale pin under it the GPU loses the mapping: "Memory access fault by GPU ... on address <a heap page>", abort() — seen four
# times over three rounds, always in the first small `.cpu()` after the multi-megabyte copies of the prover-shaped test
# (profiles/r06/sigabrt_recurrence.txt).  The library itself hands the runtime no pageable memory (ctx.hpp: staging rings);
# this keeps torch's copies IN THE TEST PROCESS off the pinning path as well: below this size (MiB) the runtime stages a
# pageable copy through its own buffers.  Must be set before the HIP runtime is loaded (i.e. before torch is imported).
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1048576")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

def _load_abort_trace():
    """tests/tools/abrt_trace.c: if this process ever dies of SIGABRT / SIGSEGV / SIGBUS, the raising thread's native
    frames and the tail of the captured stderr reach the real stderr (round 4 lost two suite runs to a bare SIGABRT
    from a runtime thread that left nothing to read).  A signal handler only: nothing runs until the signal does.
    Loaded here — an initial conftest is imported before pytest enables faulthandler, which then chains to it.  pytest's
    fd capture is already in place by then, so what the handler writes to "stderr" is captured too: its real output is
    the file gpurun_out/abort_traces/abort_trace.<pid>.log, created when the signal arrives."""
    import ctypes
    import subprocess
    src = os.path.join(ROOT, "tests", "tools", "abrt_trace.c")
    lib = os.path.join(ROOT, "tests", "tools", "libabrt_trace.so")
    try:
        if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O1", "-g", "-fPIC", "-shared", src, "-o", lib],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        trace_dir = os.path.join(ROOT, "gpurun_out", "abort_traces")      # gpurun merges gpurun_out/ back
        os.makedirs(trace_dir, exist_ok=True)
        os.environ.setdefault("HODOR_ABORT_TRACE_DIR", trace_dir)
        ctypes.CDLL(lib)
    except Exception:
        pass      # a diagnostic aid: never a reason not to run the tests


if os.environ.get("HODOR_TEST_ABORT_TRACE", "1") != "0":
    _load_abort_trace()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


FIELDS = {
    "bn256": (52435875175126190479447740508185965837690552500527637822603658699938581184513, 7),
    "experiments": (3618502788666131213697322783095070105623107215331596699973092056135872020481, 3),
    # not a field of the reference: the BN254 scalar field (254 bits, 2-adicity 28, p mod 2^29 != 1) keeps
    # the "modulus is a run-time parameter" claim honest — both reference fields have S >= 32
    "bn254": (21888242871839275222246405745257275088548364400416034343698204186575808495617, 5),
}


@pytest.fixture(scope="session", params=["bn256", "experiments", "bn254"])
def field_name(request):
    return request.param


@pytest.fixture(scope="session")
def oracles():
    from oracle.oracle import Oracle
    return {k: Oracle(*v) for k, v in FIELDS.items()}


@pytest.fixture(scope="session")
def gpu_ctxs():
    """One hodor_ctx per field on cuda:0.  Fails loudly when the HIP extension is missing."""
    import hodor_amd
    ctxs = {k: hodor_amd.Context(v[0], v[1], device=0) for k, v in FIELDS.items()}
    yield ctxs
    for c in ctxs.values():
        c.close()


def variant_lib(target, name):
    """hodor_amd/<name>, a build variant of the library (`make <target>` in hodor_amd/csrc: nolate, bounds, ...), rebuilt when
    it is missing or does not export the binding's whole symbol list — a variant left over from an older tree would fail to
    load (or load and test yesterday's code).  Returns the path, or None when it cannot be built here."""
    import subprocess
    lib = os.path.join(ROOT, "hodor_amd", name)
    probe = ("import ctypes, sys; sys.path.insert(0, %r); from hodor_amd import _lib; h = ctypes.CDLL(%r); "
             "sys.exit(0 if all(hasattr(h, s) for s in _lib.EXPORTS) else 1)" % (ROOT, lib))

    def current():     # in a process of its own: two copies of the kernels must not meet in this one
        return os.path.exists(lib) and subprocess.run([sys.executable, "-c", probe], capture_output=True).returncode == 0
    if not current():
        subprocess.run(["make", "-C", os.path.join(ROOT, "hodor_amd", "csrc"), target], capture_output=True)
    return lib if current() else None


def need_hbm(bytes_needed, what):
    """Flagship-size GPU tests must not turn into an unread `s`: free the caching allocator and retry once; skip only on
    a part whose TOTAL memory cannot hold the case (an MI355X has 288 GB), FAIL when a big-enough part is merely full."""
    import gc

    import torch
    for _ in range(2):
        free, total = torch.cuda.mem_get_info()
        if free >= bytes_needed:
            return
        gc.collect()
        torch.cuda.empty_cache()
        from hodor_amd import _lib
        _lib.trim_all()            # the contexts' device pools (FRI prototypes, handles) too
        torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if free >= bytes_needed:
        return
    if total < bytes_needed * 1.1:
        pytest.skip("%s: the device has %.0f GiB in total, the case needs %.0f GiB" % (what, total / 2**30, bytes_needed / 2**30))
    pytest.fail("%s: only %.0f of %.0f GiB of HBM are free (needs %.0f GiB) after emptying the allocator cache — "
                "something earlier in the run holds device memory" % (what, free / 2**30, total / 2**30, bytes_needed / 2**30))
