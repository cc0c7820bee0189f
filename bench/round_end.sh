#!/bin/bash
# Everything a round's evidence consists of, in one gpurun call: rocprofv3 kernel trace + PMC passes of the bench command
# (bench/profile.sh), the default bench line, the prove-shaped run from C++ and from Python, the world-1 transport table,
# the north-star size sweep.  Outputs under gpurun_out/<tag>/ (copy what is to be judged into profiles/<tag>/).
# usage: bash bench/round_end.sh r06
TAG=${1:-r06}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
bash bench/profile.sh $TAG > "$OUT/profile.log" 2>&1
# the bench line quotes counter traffic only from a profile of THIS build: make the one just taken the committed one's stand-in
mkdir -p "$ROOT/profiles/$TAG" && cp "$ROOT/gpurun_out/prof_$TAG/pmc_traffic.json" "$ROOT/profiles/$TAG/pmc_traffic.json" 2>/dev/null
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
bash bench/prove_shape_cpp.sh "$OUT/prove_shape_cpp.txt"
python bench/selftest_cost.py > "$OUT/selftest.txt" 2>&1
bash bench/direct_ab.sh 2 > "$OUT/direct_transport_world1.txt" 2>&1
python bench/size_sweep.py --out "$OUT/size_sweep.json" > "$OUT/size_sweep.log" 2>&1
tail -c 1500 "$OUT/bench_default.json"; tail -20 "$OUT/direct_transport_world1.txt"; head -12 "$OUT/prove_shape_cpp.txt" | cut -c1-300
