#!/bin/bash
# Which runtime calls does one prove-shaped run make?  `rocprofv3 --hip-trace --stats` of tests/host_cpp/prove_shape.cpp
# (C++ through hodor.hpp only; 5 proofs after one warm-up proof) — the pool's claim is that no handle `free` waits for the
# device: hipDeviceSynchronize must not appear outside context creation / destruction, and hipMalloc / hipFree must not
# grow with the number of proofs once the pool is warm.  Run twice (2 and 6 proofs) so that per-proof counts are a difference.
# usage: bash bench/hip_trace_proof.sh <out.txt>
OUT=${1:-/dev/stdout}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
EXE=/tmp/prove_shape_cpp
g++ -O2 -std=c++17 -pthread "$ROOT/tests/host_cpp/prove_shape.cpp" -L"$ROOT/hodor_amd" -lhodor_gpu -Wl,-rpath,"$ROOT/hodor_amd" -o $EXE || exit 9
export TMPDIR=/tmp
{
  for reps in 2 6; do
    D=/tmp/hiptrace_$reps
    rm -rf $D
    (cd /tmp && rocprofv3 --hip-trace --stats --output-format csv -d $D -o t -- $EXE 20 4 16 0 /tmp/proof_trace.bin $reps 0 1 > $D.log 2>&1)
    echo "== $reps proofs (+ 1 warm-up): $(grep -m1 total_ms $D.log | grep -o '"total_ms": [0-9.]*') under the tracer"
    python3 - "$D" <<'PY'
import csv, glob, sys
rows = {}
files = [f for f in glob.glob(sys.argv[1] + "/**/*stats.csv", recursive=True) if "hip" in f.rsplit("/", 1)[-1]]
if not files:
    print("  no hip stats file among", glob.glob(sys.argv[1] + "/**/*", recursive=True)[:20])
for f in files:
    for r in csv.DictReader(open(f)):
        rows[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6)
keep = ("hipDeviceSynchronize", "hipStreamSynchronize", "hipEventSynchronize", "hipMalloc", "hipFree", "hipHostMalloc",
        "hipHostFree", "hipMemcpyAsync", "hipMemcpy", "hipLaunchKernel", "hipModuleLaunchKernel", "hipExtLaunchKernel",
        "hipEventRecord", "hipStreamWaitEvent", "hipMemsetAsync", "hipEventQuery", "hipEventCreateWithFlags")
for k in keep:
    if k in rows:
        print("  %-26s calls %7d   total %10.3f ms" % (k, rows[k][0], rows[k][1]))
other = sorted((k for k in rows if k not in keep), key=lambda k: -rows[k][0])[:8]
print("  others by calls: " + ", ".join("%s %d" % (k, rows[k][0]) for k in other))
PY
  done
} > "$OUT" 2>&1
