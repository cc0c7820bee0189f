#!/usr/bin/env python3
"""North-star size sweep (BASELINE.json: "throughput on synthetic 2^20-2^30 domains ... as absolute numbers
and as achieved fraction of HBM peak"): NTT, iNTT, LDE x8 + Merkle commit, FRI commit on one MI355X.

    python bench/size_sweep.py [--out gpurun_out/size_sweep.json] [--max-log-n 30]

Per size: ms, field elements/s, algorithmic GB/s and its fraction of the 8 TB/s HBM peak (SURVEY.md
§8(d): a transform moves n x 32 B x 2; LDE+commit n x 32 + 2 x (8n x 32); FRI commit ~6 x n x 32), and for the
transforms the v_mad_u64_u32 rate against the measured 30 Tmad/s issue ceiling.  Inputs are the
SplitMix64 stream; every transform is checked by its inverse, and where tests/golden/fullsize_digests.json
holds the CPU oracle's answer (NTT 2^20/22/24, LDE 2^18/22, FRI 2^20/26) the outputs are compared with it."""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import hodor_amd  # noqa: E402

HBM_PEAK, MAD_PEAK = 8000.0, 30.0
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")))


def digest(t):
    return hashlib.blake2s(memoryview(t.cpu().numpy()).cast("B"), digest_size=32).hexdigest()


def timeit(f, min_ms=150.0, max_reps=200):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < min_ms / 2:    # warm the clocks
        f()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps, total = 0, 0.0
    while total < min_ms and reps < max_reps:
        e0.record()
        f()
        e1.record()
        torch.cuda.synchronize()
        total += e0.elapsed_time(e1)
        reps += 1
    return total / reps


def plan(log_n):
    passes = max(1, -(-log_n // 9)) if log_n > 10 else 1
    base, rem = divmod(log_n, passes)
    radices = [base + (1 if i < rem else 0) for i in range(passes)]
    products = sum(0.5 * r - (0.75 if r % 2 == 0 else 0.5) for r in radices) + sum(min(i, 2) for i in range(passes))
    return passes, products


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "size_sweep.json"))
    ap.add_argument("--min-log-n", type=int, default=20)
    ap.add_argument("--max-log-n", type=int, default=30)
    args = ap.parse_args()
    knobs = {k: v for k, v in os.environ.items() if k.startswith("HODOR_")}
    ctx = hodor_amd.Context(hodor_amd.BN256_FR_MODULUS, hodor_amd.BN256_FR_GENERATOR, device=0)
    free, _ = torch.cuda.mem_get_info()
    rows = []
    for log_n in range(args.min_log_n, args.max_log_n + 1):
        n = 1 << log_n
        row = {"log_n": log_n}
        if free < 3.4 * n * 32:
            row["skipped"] = "not enough free HBM"
            rows.append(row)
            continue
        seed = FIX["ntt"].get(str(log_n), {"seed": 0x484F444F52})["seed"]
        a = torch.empty((n, 4), dtype=torch.int64, device="cuda")
        ctx.gen_elements_dev(a, 0, n, seed)
        b, c = torch.empty_like(a), torch.empty_like(a)
        passes, products = plan(log_n)
        for name, fn in (("ntt", lambda: ctx.poly_fft_dev(a, b, log_n)), ("intt", lambda: ctx.poly_ifft_dev(b, c, log_n))):
            ms = timeit(fn)
            gbs = 2.0 * n * 32 / (ms * 1e-3) / 1e9
            mads = n * (products * 108 + 9 * passes) / (ms * 1e-3) / 1e12
            row[name] = {"ms": ms, "elems_per_s": n / (ms * 1e-3), "alg_gb_per_s": gbs, "hbm_frac": gbs / HBM_PEAK,
                         "passes": passes, "products_per_element": products, "tmad_per_s": mads,
                         "mad_frac": mads / MAD_PEAK}
        torch.cuda.synchronize()
        assert torch.equal(a, c), "iNTT(NTT(x)) != x at 2^%d" % log_n
        row["roundtrip"] = True
        if str(log_n) in FIX["ntt"]:
            assert digest(b) == FIX["ntt"][str(log_n)]["fft"], "NTT differs from the CPU oracle at 2^%d" % log_n
            row["ntt_equals_cpu_oracle_digest"] = True
        del b, c

        # LDE x8 + commit and FRI commit ON this domain size: coefficients n/8, codeword n
        f = 8
        log_deg = log_n - 3
        fx = FIX["lde"].get(str(log_deg))
        coeffs = torch.empty((1 << log_deg, 4), dtype=torch.int64, device="cuda")
        ctx.gen_elements_dev(coeffs, 0, 1 << log_deg, fx["seed"] if fx else 0x484F444F53)
        nodes = torch.empty((n, 32), dtype=torch.uint8, device="cuda")

        def lde_commit():
            ctx.poly_lde_dev(coeffs, a, log_deg, f)
            ctx.iop_create_dev(a, n, nodes)
        ms = timeit(lde_commit)
        alg = (n / f) * 32 + 2.0 * n * 32
        row["lde8_commit"] = {"ms": ms, "coeffs_log": log_deg, "gib_per_s": alg / 2**30 / (ms * 1e-3),
                              "hbm_frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK, "root": bytes(nodes[1].cpu().numpy()).hex()}
        if fx:
            assert row["lde8_commit"]["root"] == fx["root"], "LDE+commit root differs from the CPU oracle"
            row["lde8_commit"]["root_equals_cpu_oracle"] = True
        del nodes
        fx = FIX["fri"].get(str(log_n))
        if fx:
            ctx.gen_elements_dev(coeffs, 0, 1 << log_deg, fx["seed"])
            ctx.poly_lde_dev(coeffs, a, log_deg, f)
        torch.cuda.synchronize()
        proto = ctx.fri_commit_dev(a, n, f, 1)
        ser = proto.serialized
        rounds = proto.num_steps
        proto.free()
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            t = time.perf_counter()
            p = ctx.fri_commit_dev(a, n, f, 1)
            best = min(best, time.perf_counter() - t)
            p.free()
        ms = best * 1e3
        row["fri_commit"] = {"ms": ms, "rounds": rounds, "gib_per_s": 6.0 * n * 32 / 2**30 / (ms * 1e-3),
                             "hbm_frac": 6.0 * n * 32 / (ms * 1e-3) / 1e9 / HBM_PEAK}
        if fx:
            assert ser.hex() == fx["serialized"], "FRI prototype differs from the CPU oracle"
            row["fri_commit"]["bytes_equal_cpu_oracle"] = True
        # the same codeword in the COSET2 tree format (opt-in, HODOR_COMBINER_COSET2): commit and FRI commit
        c2 = hodor_amd.COSET2
        nodes2 = torch.empty((n // 2, 32), dtype=torch.uint8, device="cuda")
        ms = timeit(lambda: ctx.iop_create_combined_dev(a, n, c2, nodes2))
        row["commit_coset2_ms"] = ms
        del nodes2
        proto = ctx.fri_commit_dev(a, n, f, 1, combiner=c2)
        ser2 = proto.serialized
        proto.free()
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            t = time.perf_counter()
            p = ctx.fri_commit_dev(a, n, f, 1, combiner=c2)
            best = min(best, time.perf_counter() - t)
            p.free()
        row["fri_commit_coset2"] = {"ms": best * 1e3, "speedup_vs_reference_format": row["fri_commit"]["ms"] / (best * 1e3)}
        fx2 = FIX.get("fri_coset2", {}).get(str(log_n))
        if fx and fx2:
            assert ser2.hex() == fx2["serialized"], "COSET2 FRI prototype differs from the CPU oracle"
            row["fri_commit_coset2"]["bytes_equal_cpu_oracle"] = True
        del a, coeffs
        torch.cuda.empty_cache()
        rows.append(row)
        print(json.dumps(row), flush=True)
    out = {"device": torch.cuda.get_device_name(0), "field": "src/bn256.rs Fr", "hbm_peak_gb_per_s": HBM_PEAK,
           "mad_peak_tmad_per_s": MAD_PEAK, "knobs": knobs, "rows": rows}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
