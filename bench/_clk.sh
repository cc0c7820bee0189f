REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_clk
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FULL="python $REPO/bench.py --no-cpu-baseline --steps 20 --warmup 5"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/clk -o p -- $FULL > $OUT/clk.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/sq2 -o p -- $FULL > $OUT/sq2.log 2>&1
python - <<PY
import csv, glob, collections
for sub in ("clk","sq2"):
  for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][-44:], r["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
        if "Start_Timestamp" in r: agg[k][2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, v in sorted(agg.items()):
        if any(t in k[0] for t in ("ntt_pass", "merkle", "fri")):
            print(sub, k[0], k[1], "dispatches", v[0], "avg", v[1] / v[0], "avg_ns", v[2] / v[0], "per_ns", (v[1]/v[2]) if v[2] else None)
PY
tail -3 $OUT/clk.log $OUT/sq2.log
