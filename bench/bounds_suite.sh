#!/bin/bash
# The whole GPU suite — Python tests, the C / C++ programs they build, the plan fuzzer, the two-process transport soak —
# on the BOUNDS-CHECKED build of the library (hodor_amd/csrc/bounds.cuh): every global access and LDS slot of every kernel
# against the extents its launcher declares.  One line per process goes to $OUT/bounds_report.txt (checked launches,
# device-side and host-side hits); a hit also fails the call that meets it (HODOR_ERR_DEVICE), hence the test.
# usage: bash bench/bounds_suite.sh [outdir=gpurun_out/bounds]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$ROOT/gpurun_out/bounds}
mkdir -p "$OUT"
cd "$ROOT"
[ -f hodor_amd/libhodor_gpu_bounds.so ] || make -C hodor_amd/csrc bounds > "$OUT/build.log" 2>&1 || { echo "bounds build failed"; exit 9; }
rm -f "$OUT/bounds_report.txt"
export HODOR_LIB=$ROOT/hodor_amd/libhodor_gpu_bounds.so HODOR_BOUNDS_REPORT=$OUT/bounds_report.txt HODOR_SUITE_LIB=1
# (tests that A/B other twin builds through HODOR_LIB of their own — nolate — and the bounds build's own tests keep their library)
( time python -m pytest tests -m gpu -q -p no:cacheprovider ) > "$OUT/suite.log" 2>&1
tail -5 "$OUT/suite.log"
python3 - "$OUT/bounds_report.txt" <<'PY'
import re, sys
procs = launches = dev = host = 0
firsts = []
for line in open(sys.argv[1]):
    m = re.match(r"pid \d+: (\d+) checked launches, (\d+) device-side hits, (\d+) host-side hits(.*)", line)
    if not m: continue
    procs += 1; launches += int(m.group(1)); dev += int(m.group(2)); host += int(m.group(3))
    if m.group(4).strip(): firsts.append(m.group(4).strip())
print("bounds build: %d processes, %d checked kernel launches, %d device-side hits, %d host-side hits" % (procs, launches, dev, host))
for f in firsts[:10]: print("  ", f)
PY
