#!/bin/bash
# Where one prove-shaped run spends its device time, by kernel: `rocprofv3 --kernel-trace --stats` of
# tests/host_cpp/prove_shape.cpp (device-resident from_arp, 5 proofs + 1 warm-up), per-proof milliseconds per kernel.
# usage: bash bench/kernel_trace_proof.sh <out.txt>
OUT=${1:-/dev/stdout}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
EXE=/tmp/prove_shape_cpp
g++ -O2 -std=c++17 -pthread "$ROOT/tests/host_cpp/prove_shape.cpp" -L"$ROOT/hodor_amd" -lhodor_gpu -Wl,-rpath,"$ROOT/hodor_amd" -o $EXE || exit 9
export TMPDIR=/tmp
D=/tmp/ktrace_proof
rm -rf $D
(cd /tmp && HODOR_SELFTEST=0 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- $EXE 20 4 16 0 /tmp/proof_trace.bin 5 0 1 > $D.log 2>&1)
{
  grep -o '"total_ms": [0-9.]*' $D.log | head -1
  python3 - "$D" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6))
rows.sort(key=lambda r: -r[2])
proofs = 6.0   # 5 timed + 1 warm-up (from_arp's precompute runs twice more outside the proofs: its kernels are in the totals)
tot = sum(r[2] for r in rows)
print("all kernels: %.2f ms per proof over %d proofs" % (tot / proofs, proofs))
for name, calls, ms in rows[:28]:
    print("  %8.3f ms/proof  %6.1f launches/proof  %s" % (ms / proofs, calls / proofs, name[:110]))
PY
} > "$OUT" 2>&1
