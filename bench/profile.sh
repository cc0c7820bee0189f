#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's kernels.  Run on the GPU box from the repo root:
#   bash bench/profile.sh <tag>      (outputs under gpurun_out/prof_<tag>/, summary in summary.txt)
# Counters are collected in their own passes (--kernel-trace + --pmc only), one --pmc group per run.
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
NTT="python $REPO/bench.py --no-cpu-baseline --no-extra"   # default steps/warmup: clocks need ~100 ms to ramp
FULL="python $REPO/bench.py --no-cpu-baseline --steps 5 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o ntt -- $NTT > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_full -o full -- $FULL > $OUT/trace_full.log 2>&1
pmc() {  # name, command, counters...
    local name=$1; shift; local cmd=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- $cmd > $OUT/$name.log 2>&1
}
pmc pmc_fetch "$NTT" FETCH_SIZE
pmc pmc_write "$NTT" WRITE_SIZE
pmc pmc_sq "$NTT" SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pmc pmc_lds "$NTT" SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pmc pmc_l2 "$NTT" TCC_HIT_sum TCC_MISS_sum
# Merkle / FRI kernels (full bench: LDE x8 + commit, FRI commit 2^26)
pmc full_fetch "$FULL" FETCH_SIZE
pmc full_write "$FULL" WRITE_SIZE
pmc full_sq "$FULL" SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
# the two `extra` workloads alone, a known number of times each: HBM traffic per run for the second headline metric
EXTRA_RUNS=4
export HODOR_SELFTEST=0     # (the context's start-up self-test launches a few of the same kernels: not part of a run)
for w in lde_commit fri_commit; do
  pmc extra_${w}_fetch "python $REPO/bench/extra_workload.py $w $EXTRA_RUNS" FETCH_SIZE
  pmc extra_${w}_write "python $REPO/bench/extra_workload.py $w $EXTRA_RUNS" WRITE_SIZE
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/extra_${w}_trace -o t -- python $REPO/bench/extra_workload.py $w $EXTRA_RUNS > $OUT/extra_${w}_trace.log 2>&1
done
unset HODOR_SELFTEST
python - > $OUT/summary.txt <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, "$REPO")
def stats(sub):
    for f in glob.glob("$OUT/%s/**/*kernel_stats.csv" % sub, recursive=True):
        print("==", sub, f.split("/")[-1])
        for r in list(csv.DictReader(open(f)))[:14]:
            print("  %-60s calls %6s avg_ns %12s total_ns %14s pct %s" % (r["Name"][:60], r["Calls"], r["AverageNs"], r["TotalDurationNs"], r["Percentage"]))
stats("trace"); stats("trace_full")
# per-pass split of the NTT + iNTT step: the k_ntt_pass dispatches repeat with period 6 (3 forward, 3 inverse)
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_ntt_pass" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) % 6:]          # drop table-building stragglers at the front, keep whole steps
    rows = rows[len(rows) // 2 // 6 * 6:]   # second half: clocks settled
    per = collections.defaultdict(list)
    for i, r in enumerate(rows):
        per[i % 6].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in sorted(per):
        print("k_ntt_pass dispatch %d of 6 (%s pass %d): avg %.1f us over %d" % (k, "forward" if k < 3 else "inverse", k % 3 + 1, sum(per[k]) / len(per[k]) / 1e3, len(per[k])))
for sub in ("pmc_fetch","pmc_write","pmc_sq","pmc_lds","pmc_l2","full_fetch","full_write","full_sq"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].split("(")[0][-44:], r["Counter_Name"])
            agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
        for k, v in sorted(agg.items()):
            if any(t in k[0] for t in ("ntt_pass", "merkle", "fri_fold", "fri_tail", "fri_round")):
                print(sub, k[0], k[1], "dispatches", v[0], "avg", v[1] / v[0])
# pmc_traffic.json: what bench.py quotes as roofline.traffic / rocprofv3_avg_launch_ms, stamped with the sha256 of
# the kernel's sources so that a profile of another build can never travel in a fresh bench line
def avg_counter(sub, name):
    vals = []
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_ntt_pass<0" in r["Kernel_Name"] and r["Counter_Name"] == name:
                vals.append(float(r["Counter_Value"]))
    vals = vals[len(vals) // 2:]          # second half: tables built, clocks settled
    return sum(vals) / len(vals) if vals else None
fetch_kb, write_kb = avg_counter("pmc_fetch", "FETCH_SIZE"), avg_counter("pmc_write", "WRITE_SIZE")
trace_ms = None
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_ntt_pass<0" in r["Name"]:      # the plain instantiation: the bench's kernel (<1, .> is the start-up self-test's 4-step probe)
            trace_ms = float(r["AverageNs"]) / 1e6
if fetch_kb and write_kb:
    import bench
    doc = {"kernel": "k_ntt_pass<0>", "log_n": 24,
           "command": "python bench.py --no-cpu-baseline --no-extra (bench/profile.sh $TAG)",
           "sources_sha256": bench.kernel_sources_sha256(),
           "fetch_size_kb_avg": fetch_kb, "fetch_bytes_corrected_x2": fetch_kb * 1024 * 2,
           "write_size_kb_avg": write_kb, "write_bytes": write_kb * 1024,
           "hbm_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024,
           "kernel_trace_avg_launch_ms": trace_ms,
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); counters from "
                   "separate --pmc passes; averages over the second half of the dispatches"}
    # the extras: bytes per RUN of the whole workload (all dispatches of the kernels that belong to it, / the number of runs)
    def per_run(workload, runs=$EXTRA_RUNS):
        # lde_commit = the LDE's passes + the tree; fri_commit = trees, folds, challenges (its k_ntt_pass launches are the
        # codeword's LDE, made ONCE before the runs, and the final 2-point transform: not the commit's traffic)
        keep = ("ntt_pass", "merkle") if workload == "lde_commit" else ("merkle", "fri_fold", "fri_tail", "fri_round", "k_challenge")
        res, by_kernel = {}, collections.defaultdict(lambda: [0.0, 0.0, 0])
        for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            total = 0.0
            for f in glob.glob("$OUT/extra_%s_%s/**/*counter_collection.csv" % (workload, kind), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] != counter or not any(k in r["Kernel_Name"] for k in keep): continue
                    v = float(r["Counter_Value"]) * 1024 * (2 if kind == "fetch" else 1)
                    total += v
                    name = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
                    by_kernel[name][0 if kind == "fetch" else 1] += v
                    if kind == "fetch": by_kernel[name][2] += 1
            res[kind + "_bytes_per_run"] = total / runs if total else None
        if res.get("fetch_bytes_per_run") and res.get("write_bytes_per_run"):
            res["hbm_bytes_per_run"] = res["fetch_bytes_per_run"] + res["write_bytes_per_run"]
        res["runs"] = runs
        res["by_kernel_per_run"] = {k: {"fetch_bytes": v[0] / runs, "write_bytes": v[1] / runs, "dispatches": v[2] / runs} for k, v in sorted(by_kernel.items())}
        ms = {}
        for f in glob.glob("$OUT/extra_%s_trace/**/*kernel_stats.csv" % workload, recursive=True):
            for r in csv.DictReader(open(f)):
                if any(k in r["Name"] for k in keep):
                    ms[r["Name"].split("(")[0].split("::")[-1][:40]] = float(r["TotalDurationNs"]) / 1e6 / runs
        res["kernel_ms_per_run"] = ms
        return res
    doc["extras"] = {"sources_sha256": bench.extras_sources_sha256(), "command": "python bench/extra_workload.py <workload> $EXTRA_RUNS",
                     "lde_commit": per_run("lde_commit"), "fri_commit": per_run("fri_commit")}
    json.dump(doc, open("$OUT/pmc_traffic.json", "w"), indent=1)
    print("pmc_traffic.json:", json.dumps(doc))
PY
cat $OUT/summary.txt
# the raw per-dispatch tables are tens of MiB (gpurun merges at most 64 MiB back): keep the per-kernel statistics, the
# summary and pmc_traffic.json unless KEEP_RAW=1
if [ "${KEEP_RAW:-0}" != "1" ]; then
  find $OUT -name '*kernel_trace.csv' -delete
  find $OUT -name '*counter_collection.csv' -delete
  find $OUT -name '*agent_info.csv' -delete
fi
du -sh $OUT
