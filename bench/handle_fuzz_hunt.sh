#!/bin/bash
# A longer hunt with tests/test_gpu_handle_fuzz.py than the suite's 24 programs: <programs> seeds per field of <steps> steps on
# the shipped build, then on the bounds-checked build (every launch of every program checked against the extents its launcher
# declares and the pool's allocation registry).
# usage: bash bench/handle_fuzz_hunt.sh <out_dir> [programs=150] [steps=200] [seed0=100]
OUT=${1:-gpurun_out/fuzz_hunt}; N=${2:-150}; STEPS=${3:-200}; SEED0=${4:-100}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd "$ROOT"
export HODOR_FUZZ_PROGRAMS=$N HODOR_FUZZ_STEPS=$STEPS HODOR_FUZZ_SEED0=$SEED0
(time python -m pytest tests/test_gpu_handle_fuzz.py -q -x) > "$OUT/shipped.log" 2>&1
tail -6 "$OUT/shipped.log" | cut -c1-300
HODOR_BOUNDS_REPORT="$OUT/bounds_report.txt" HODOR_LIB="$ROOT/hodor_amd/libhodor_gpu_bounds.so" \
  python -m pytest tests/test_gpu_handle_fuzz.py -q -x > "$OUT/bounds.log" 2>&1
tail -4 "$OUT/bounds.log" | cut -c1-300
cat "$OUT/bounds_report.txt"
