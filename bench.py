#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native hodor hot path.

Metric (BASELINE.json): NTT field-elements/s on the 2^24 domain over the src/bn256.rs field
(config[1]: "2^24-point NTT + iNTT on 1x MI355X"), with the LDE x8 + Merkle-commit GiB/s
(config[2]) reported beside it in `extra`.

A "step" = one forward NTT followed by one inverse NTT (Polynomial::fft + Polynomial::ifft,
/root/reference/src/polynomials/mod.rs:611-624, :773-798) of a device-resident 2^24-element
polynomial; inputs are resident in HBM before the timed region.  With N > 1 ranks every rank
transforms its own polynomial (the prover holds one per register, src/prover/mod.rs:73-76): weak
scaling, no data-path collective.  `value` = field elements transformed by all ranks / max-over-ranks
time.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MAD_PEAK_TOPS = 30.0     # v_mad_u64_u32 lane-ops/s, 256 CUs x 48.8 per clock x 2.4 GHz (profiles/r01/microbench_gfx950.txt)
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
LOG_N = 24                # BASELINE.json config[1]
LDE_LOG_N, LDE_FACTOR = 22, 8   # BASELINE.json config[2]


def cpu_baseline(seconds_budget=20.0):
    """Reference CPU path (restated, oracle/hodor_oracle.c: best_fft -> parallel_fft,
    /root/reference/src/fft/fft.rs:5-124) on the host cores: config[0], 2^20-point NTT."""
    from oracle import pyref as P
    from oracle.oracle import Oracle
    O = Oracle(P.BN256.p, P.BN256.g)
    log_n = 20
    n = 1 << log_n
    a = O.random_elements(n, 2024)
    _, k, w = O.domain(n)
    reps, total = 0, 0.0
    while reps < 1 or (total < seconds_budget / 2 and reps < 8):
        b = a.copy()
        t = time.perf_counter()
        O.best_fft(b, w, k)
        total += time.perf_counter() - t
        reps += 1
    return {"value": n * reps / total, "unit": "field-elems/s", "cores": O.cpus, "kind": "port",
            "sample": "%d x 2^20-point NTT (config[0]) via the restated Worker/parallel_fft schedule, %.1f s"
                      % (reps, total)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # ~0.4 s timed: long enough for the clocks to settle
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--log-n", type=int, default=LOG_N)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--mode", choices=["replicas", "sixstep"], default="replicas",
                    help="N > 1: 'replicas' = one independent 2^log_n polynomial per GPU (default); "
                         "'sixstep' = ONE transform of 2^log_n * N points split over the ranks, transposes as "
                         "RCCL all-to-alls (hodor_amd/sixstep.py, BASELINE config[4] shape)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import hodor_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    if world > 1 or "RANK" in os.environ:
        # RCCL prints banner lines ("Hostname : ...", "Librccl path : ...") on STDOUT when the first
        # communicator is created; keep stdout to the single JSON line by parking fd 1 meanwhile.
        sys.stdout.flush()
        saved = os.dup(1)
        devnull = os.open(os.devnull, os.O_WRONLY)
        os.dup2(devnull, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            warm = torch.zeros(1, device="cuda")
            dist.all_reduce(warm)
            torch.cuda.synchronize()
        finally:
            os.dup2(saved, 1)
            os.close(saved)
            os.close(devnull)

    ctx = hodor_amd.Context(hodor_amd.BN256_FR_MODULUS, hodor_amd.BN256_FR_GENERATOR, device=local_rank)
    log_n = args.log_n
    n = 1 << log_n

    # synthetic input: random field elements (Montgomery images), generated on the device
    a = random_elements(torch, n, 0x484F444F52 + rank)
    b = torch.empty_like(a)
    c = torch.empty_like(a)
    # a non-default stream: its handle is non-null, so the library launches on it (NULL would select
    # the context's own stream) and the torch events below bracket exactly these kernels
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(side)
    stream = side.cuda_stream

    if args.mode == "sixstep":
        from hodor_amd.sixstep import HipBackend, sixstep_intt, sixstep_ntt
        log_total = log_n + (world.bit_length() - 1)
        assert 1 << (world.bit_length() - 1) == world, "sixstep needs a power-of-two world size"
        omega = ctx.domain(1 << log_total)[2]
        be = HipBackend(ctx, stream=stream)
        holder = {}

        def step():
            y = sixstep_ntt(be, a, log_total, omega, rank, world)
            holder["c"] = sixstep_intt(be, y, log_total, omega, rank, world)
    else:
        def step():
            ctx.poly_fft_dev(a, b, log_n, stream=stream)
            ctx.poly_ifft_dev(b, c, log_n, stream=stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if args.mode == "sixstep":
        c = holder["c"]
    if not torch.equal(a, c) and not os.environ.get("HODOR_DBG"):   # HODOR_DBG: profiling ablations only
        raise SystemExit("iNTT(NTT(x)) != x — refusing to report a number")

    def barrier():
        if world > 1:
            dist.barrier()

    barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1)          # HIP events on the launch stream
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    elems = 2.0 * n * args.steps * world       # forward + inverse
    result = {
        "metric": "ntt_field_elems_per_sec",
        "value": elems / dt,
        "unit": "field-elems/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": "2^%d-point NTT + iNTT over the src/bn256.rs Fr field, device-resident, "
                               "bit-exact vs CPU oracle (BASELINE config[1])" % log_n,
                   "log_n": log_n, "field": "bn256.rs Fr (255-bit, R=2^256)",
                   "arithmetic": "exact integer: 256-bit Montgomery elements as 9 x 29-bit limbs in u32, "
                                 "32x32->64 multiply-accumulate (v_mad_u64_u32)",
                   "parallelism": ("6-step, 2^%d points over %d GPUs, RCCL all-to-all transposes"
                                   % (log_n + world.bit_length() - 1, world)) if args.mode == "sixstep"
                   else ("1 polynomial per GPU" if world > 1 else "1 GPU")},
    }

    if rank == 0:
        # roofline of the dominant kernel (k_ntt_pass): one transform = `passes` launches and must
        # move 2 x n x 32 B at least once; each launch is charged 1/passes of that.
        passes = max(1, -(-log_n // 9)) if log_n > 10 else 1     # plan_radices() in csrc/abi.hip
        launches = 2 * passes * args.steps
        avg_launch_ms = kernel_ms / launches
        alg_bytes_per_launch = 2.0 * n * 32 / passes
        achieved = alg_bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
        traffic = profiled_ms = None   # from the committed rocprofv3 passes of the same command (not live)
        try:
            pmcs = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.startswith("r"))
            t = json.load(open(os.path.join(ROOT, "profiles", pmcs[-1], "pmc_traffic.json")))
            if t.get("log_n") == log_n and args.mode == "replicas":
                traffic = t["hbm_bytes_per_launch"]
                profiled_ms = t.get("kernel_trace_avg_launch_ms")
        except Exception:
            pass
        result["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                              "kernel": "k_ntt_pass", "avg_launch_ms": avg_launch_ms,
                              "rocprofv3_avg_launch_ms": profiled_ms,
                              "launches_per_transform": passes,
                              "alg_bytes_per_launch": alg_bytes_per_launch,
                              "note": "integer-ALU-bound kernel (v_mad_u64_u32); HBM fraction reported as required"}
        # the resource that actually binds: v_mad_u64_u32 issue.  Products per element and transform =
        # butterflies (0.5 per stage, minus the trivial twiddles of each pass's first two stages) +
        # inter-pass twiddles (1 for the second pass, 2 from the third on); 162 mads per 9 x 29 product,
        # 9 more per element where a pass reduces its output.  Peak = bench/microbench.hip on this part.
        base, rem = divmod(log_n, passes)
        radices = [base + (1 if i < rem else 0) for i in range(passes)]
        products = sum(0.5 * r - (0.75 if r % 2 == 0 else 0.5) for r in radices) + sum(min(i, 2) for i in range(passes))
        mads_per_launch = n * (products * 162 + 9 * passes) / passes
        result["roofline"]["valu"] = {
            "bound": "v_mad_u64_u32 issue", "products_per_element": products,
            "achieved": mads_per_launch / (avg_launch_ms * 1e-3) / 1e12, "peak": MAD_PEAK_TOPS, "unit": "Tmad/s",
            "frac": mads_per_launch / (avg_launch_ms * 1e-3) / 1e12 / MAD_PEAK_TOPS,
            "note": "mads are ~64 % of the kernel's VALU cycles; the VALU as a whole is ~92 % busy (profiles/r01/pmc_summary.md)"}
        if not args.no_extra and world == 1:
            result["extra"] = extra_lde_commit(ctx, torch, stream)
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline()
        print(json.dumps(result))
    if dist.is_initialized():
        dist.destroy_process_group()


def random_elements(torch, n, seed):
    """n random elements of the src/bn256.rs field as (n, 4) int64 limbs on the current device:
    three uniform 64-bit limbs and a top limb below floor(p / 2^224) * 2^32, hence value < p."""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    out = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    out[:, :3] = torch.randint(-2**63, 2**63 - 1, (n, 3), dtype=torch.int64, device="cuda", generator=g)
    out[:, 3] = torch.randint(0, 0x73EDA753 << 32, (n,), dtype=torch.int64, device="cuda", generator=g)
    return out


def extra_lde_commit(ctx, torch, stream):
    """config[2]: LDE x8 of a 2^22-coefficient polynomial + IOP Merkle commit, device-resident."""
    n = 1 << LDE_LOG_N
    big = n * LDE_FACTOR
    coeffs = random_elements(torch, n, 777)
    lde = torch.empty((big, 4), dtype=torch.int64, device="cuda")
    nodes = torch.empty((big, 32), dtype=torch.uint8, device="cuda")

    def run():
        ctx.poly_lde_dev(coeffs, lde, LDE_LOG_N, LDE_FACTOR, stream=stream)
        ctx.iop_create_dev(lde, big, nodes, stream=stream)

    run()
    torch.cuda.synchronize()
    reps = 5
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    lde_ms = commit_ms = 0.0
    for _ in range(reps):
        e0.record()
        ctx.poly_lde_dev(coeffs, lde, LDE_LOG_N, LDE_FACTOR, stream=stream)
        e1.record()
        ctx.iop_create_dev(lde, big, nodes, stream=stream)
        e2.record()
        torch.cuda.synchronize()
        lde_ms += e0.elapsed_time(e1)
        commit_ms += e1.elapsed_time(e2)
    lde_ms /= reps
    commit_ms /= reps
    alg_bytes = n * 32 + big * 32 + big * 32      # read coeffs + write LDE + write nodes
    out = {"workload": "LDE x8 of 2^22 + BLAKE2s Merkle commit (BASELINE config[2])",
           "lde_ms": lde_ms, "commit_ms": commit_ms,
           "lde_commit_gib_per_s": alg_bytes / 2**30 / ((lde_ms + commit_ms) * 1e-3),
           "root": bytes(nodes[1].cpu().numpy()).hex()}
    del lde, nodes, coeffs
    out["fri_commit"] = extra_fri_commit(ctx, torch, stream)
    return out


def extra_fri_commit(ctx, torch, stream):
    """config[3]: FRI commit phase on a 2^26 codeword (= LDE x8 of 2^23 random coefficients),
    lde_factor 8, final degree+1 = 1 -> 23 folding rounds; device-resident."""
    log_deg, factor = 23, 8
    n = (1 << log_deg) * factor
    coeffs = random_elements(torch, 1 << log_deg, 4242)
    code = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    ctx.poly_lde_dev(coeffs, code, log_deg, factor, stream=stream)
    torch.cuda.synchronize()
    proto = ctx.fri_commit_dev(code, n, factor, 1, stream=stream)      # warm-up: tables + slab
    first = proto.serialized
    proto.free()
    reps, total = 3, 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        proto = ctx.fri_commit_dev(code, n, factor, 1, stream=stream)  # synchronises before returning
        total += time.perf_counter() - t
        assert proto.serialized == first
        steps = proto.num_steps
        proto.free()
    ms = total / reps * 1e3
    return {"workload": "FRI commit, 2^26 codeword, lde 8, 23 rounds (BASELINE config[3])",
            "ms": ms, "rounds": steps, "gib_per_s": 6.0 * n * 32 / 2**30 / (ms * 1e-3),
            "final_root": proto.final_root.hex()}


if __name__ == "__main__":
    main()
