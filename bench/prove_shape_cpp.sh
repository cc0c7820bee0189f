#!/bin/bash
# The reference's own benchmark shape (cubic_vdf.rs:288-354: 4 registers x 2^20 rows x LDE 16) driven from C++ through
# hodor.hpp only (tests/host_cpp/prove_shape.cpp), next to the Python `_dev` replay of the same phases
# (bench/prove_shape.py): per-phase and total milliseconds, host round trips, and the proofs' digests — which must agree
# with each other (and, without --no-cpu, with the CPU port's).
# usage: bash bench/prove_shape_cpp.sh <out.txt> [log_rows=20] [registers=4] [lde_factor=16]
OUT=${1:-/dev/stdout}; LOG=${2:-20}; REGS=${3:-4}; F=${4:-16}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
EXE=/tmp/prove_shape_cpp
g++ -O2 -std=c++17 -pthread "$ROOT/tests/host_cpp/prove_shape.cpp" -L"$ROOT/hodor_amd" -lhodor_gpu -Wl,-rpath,"$ROOT/hodor_amd" -o $EXE || exit 9
{
  for comb in 0 1; do
    echo "== C++ through hodor.hpp, combiner $comb, from_arp AS WRITTEN (Polynomial::as_mut), free-running (no synchronisation between the phases)"
    $EXE $LOG $REGS $F $comb /tmp/proof_$comb.bin 5 0 0
    echo "== the same with the device drained after every phase (per-phase times)"
    $EXE $LOG $REGS $F $comb /tmp/proof_sync_$comb.bin 5 1 0
    echo "== from_arp's vectors device-resident (dense_divisor_on_coset, coset table cloned in HBM), free-running / drained"
    $EXE $LOG $REGS $F $comb /tmp/proof_res_$comb.bin 5 0 1
    $EXE $LOG $REGS $F $comb /tmp/proof_res_sync_$comb.bin 5 1 1
    cmp /tmp/proof_$comb.bin /tmp/proof_res_$comb.bin && echo "proof bytes identical in both forms of from_arp"
    echo "== A/B of hodor_fri_commit_batch_h: the same (device-resident, free-running / drained) with h1 and h2 committed one after the other"
    $EXE $LOG $REGS $F $comb /tmp/proof_seq_$comb.bin 5 0 1 0
    $EXE $LOG $REGS $F $comb /tmp/proof_seq_sync_$comb.bin 5 1 1 0
    cmp /tmp/proof_res_$comb.bin /tmp/proof_seq_$comb.bin && echo "proof bytes identical with and without the batched commit"
    python3 - <<PY
import hashlib
a=open("/tmp/proof_$comb.bin","rb").read(); b=open("/tmp/proof_sync_$comb.bin","rb").read()
print("proof %d bytes, blake2s %s%s" % (len(a), hashlib.blake2s(a, digest_size=32).hexdigest(), "" if a == b else "  (!! differs between the two runs)"))
PY
  done
  echo "== Python _dev replay (bench/prove_shape.py --no-cpu)"
  python3 "$ROOT/bench/prove_shape.py" $LOG $REGS $F --no-cpu 2>&1 | grep -v amdgpu.ids
  python3 "$ROOT/bench/prove_shape.py" $LOG $REGS $F --no-cpu --coset2 2>&1 | grep -v amdgpu.ids
} > "$OUT" 2>&1
