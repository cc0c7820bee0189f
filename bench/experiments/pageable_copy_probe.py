"""Does the HIP runtime of this torch wheel honour GPU_PINNED_MIN_XFER_SIZE (MiB: below it a pageable copy goes through the
runtime's own staging buffers instead of pinning the caller's pages)?  Times torch's pageable copies with the variable unset
and set beyond any size the suite copies.  usage: python bench/experiments/pageable_copy_probe.py"""
import os, subprocess, sys
CODE = r'''
import time, torch
for nbytes in (6 << 10, 1 << 20, 64 << 20, 512 << 20):
    d = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    h = torch.empty(nbytes, dtype=torch.uint8)
    d.cpu(); d.copy_(h); torch.cuda.synchronize()
    reps = 200 if nbytes <= (1 << 20) else 5
    t = time.perf_counter()
    for _ in range(reps): x = d.cpu()
    down = (time.perf_counter() - t) / reps
    t = time.perf_counter()
    for _ in range(reps): d.copy_(h)
    torch.cuda.synchronize()
    up = (time.perf_counter() - t) / reps
    print("  %10d bytes: download %9.3f ms (%6.2f GB/s)   upload %9.3f ms (%6.2f GB/s)" % (nbytes, down * 1e3, nbytes / down / 1e9, up * 1e3, nbytes / up / 1e9))
'''
for setting in (None, "1048576"):
    env = dict(os.environ)
    env.pop("GPU_PINNED_MIN_XFER_SIZE", None)
    if setting:
        env["GPU_PINNED_MIN_XFER_SIZE"] = setting
    print("GPU_PINNED_MIN_XFER_SIZE", setting or "<unset>")
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(out.stdout, out.stderr[-300:] if out.returncode else "")
