#!/usr/bin/env python3
"""Round-5 verdict item 7: does a radix-2^10 plan (three passes of 2^10 instead of four of 2^7..2^8 at 2^28-2^30: one pass
over HBM and one two-level inter-pass twiddle fewer, 17.5 -> ~15.5 products per element) beat the default plan where
the fourth pass starts?  The plan knobs are read once per process, so every (setting, size) is timed in a process of its
own; settings are interleaved over the rounds; every output is hashed and must agree across settings (bit-exactness).
    python bench/radix10_ab.py [--logs 27,28,29,30] [--rounds 2] [--out gpurun_out/r05/radix10_ab.txt]"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETTINGS = [
    ("default (R <= 2^9, 1024-element tiles)", {}),
    ("R 2^10, 2048-element tile (C = 2, 64-byte runs, 2 workgroups per CU)", {"HODOR_MAX_LOG_R": "10", "HODOR_TILE_LOG": "11"}),
    ("R 2^10, 1024-element tile (C = 1, 32-byte runs)", {"HODOR_MAX_LOG_R": "10", "HODOR_TILE_LOG": "10", "HODOR_MIN_LOG_C": "0"}),
    ("R 2^11, 2048-element tile (C = 1)", {"HODOR_MAX_LOG_R": "11", "HODOR_TILE_LOG": "11", "HODOR_MIN_LOG_C": "0"}),
]


def child(log_n):
    sys.path.insert(0, ROOT)
    import torch
    import hodor_amd
    ctx = hodor_amd.Context(device=0)
    n = 1 << log_n
    a = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    out = torch.empty_like(a)
    ctx.gen_elements_dev(a, 0, n, 0x484F444F52)
    ctx.poly_fft_dev(a, out, log_n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:          # warm the clocks
        ctx.poly_fft_dev(a, out, log_n)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(3, min(40, int(2000 / (2 ** (log_n - 24) * 1.7))))
    e0.record()
    for _ in range(reps):
        ctx.poly_fft_dev(a, out, log_n)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    h = hashlib.blake2s(digest_size=16)
    step = 1 << 22
    for i in range(0, n, step):                    # the whole output, 128 MiB at a time
        h.update(memoryview(out[i:i + step].cpu().numpy()).cast("B"))
    print(json.dumps({"log_n": log_n, "ms": ms, "digest": h.hexdigest(), "knobs": hodor_amd._lib.knobs_set()}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", type=int)
    ap.add_argument("--logs", default="27,28,29,30")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--out")
    args = ap.parse_args()
    if args.child is not None:
        return child(args.child)
    lines, digests = [], {}
    res = {}
    for rnd in range(args.rounds):
        for lg in [int(x) for x in args.logs.split(",")]:
            for name, env in SETTINGS:
                e = {k: v for k, v in os.environ.items() if not k.startswith("HODOR_")}
                e.update(env)
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(lg)], env=e, capture_output=True, text=True)
                js = [l for l in out.stdout.splitlines() if l.startswith("{")]
                if out.returncode or not js:
                    lines.append("2^%d  %-72s FAILED: %s" % (lg, name, (out.stderr or out.stdout)[-200:].replace("\n", " ")))
                    continue
                r = json.loads(js[-1])
                res.setdefault((lg, name), []).append(r["ms"])
                digests.setdefault(lg, set()).add(r["digest"])
    for (lg, name), ms in sorted(res.items()):
        lines.append("2^%d  %-72s %s ms  -> %.2f G elements/s" % (lg, name, " / ".join("%.2f" % m for m in ms), (1 << lg) / (min(ms) * 1e-3) / 1e9))
    for lg, ds in sorted(digests.items()):
        lines.append("2^%d  outputs of all settings bit-identical: %s" % (lg, len(ds) == 1))
    text = "\n".join(lines)
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        open(args.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
