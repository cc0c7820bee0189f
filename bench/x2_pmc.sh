#!/bin/bash
# SQ counters of k_ntt_pass for two builds of the library on one box (verdict item 1(c): the un-pinned, interleaved
# accumulator chains measured IN the kernel, with the wait counters)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04al
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in libhodor_gpu.so libhodor_gpu_x2.so; do
  HODOR_LIB=$REPO/hodor_amd/$lib rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/$lib -o p -- python $REPO/bench.py --no-cpu-baseline --no-extra --allow-knobs --soak-seconds 1 > $OUT/$lib.log 2>&1
  HODOR_LIB=$REPO/hodor_amd/$lib python $REPO/bench.py --no-cpu-baseline --no-extra --allow-knobs --soak-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', 'ms_per_step %.3f' % d['ms_per_step'])"
done
python - <<PY
import csv, glob, collections
for lib in ("libhodor_gpu.so", "libhodor_gpu_x2.so"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % lib, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_ntt_pass" in r["Kernel_Name"]:
                agg[r["Counter_Name"]][0] += 1; agg[r["Counter_Name"]][1] += float(r["Counter_Value"])
    print(lib, {k: "%.4g" % (v[1] / v[0]) for k, v in sorted(agg.items())})
PY
