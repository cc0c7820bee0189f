#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's dominant kernel.  Run on the GPU box from the repo
# root:  bash bench/profile.sh <tag>      (outputs under gpurun_out/prof_<tag>/)
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --no-cpu-baseline --no-extra"   # the default steps/warmup: clocks need ~100 ms to ramp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o ntt -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o ntt -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o ntt -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o ntt -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_lds -o ntt -- $CMD > $OUT/pmc_lds.log 2>&1
# full bench with LDE+commit for the kernel mix
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_full -o full -- python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/trace_full.log 2>&1
find $OUT -name "*.csv" | head -40
python - <<PY
import csv, glob, collections
for sub in ("pmc_fetch","pmc_write","pmc_sq","pmc_lds"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:40], r["Counter_Name"])
            agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
        for k, v in sorted(agg.items()):
            if "ntt_pass" in k[0] or "leaf" in k[0]:
                print(sub, k, "dispatches", v[0], "avg", v[1] / v[0])
PY
