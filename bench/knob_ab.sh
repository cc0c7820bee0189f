#!/bin/bash
# Same-box A/B of one tuning knob on the whole bench line (NTT step, LDE, commit, FRI), interleaved.
# usage: bash bench/knob_ab.sh HODOR_MERKLE_TAIL_LOG 5 6 7 [rounds]     (first value list, optional rounds last
#        if it is prefixed with x, e.g. x3).  Runs need --allow-knobs: bench.py refuses tuning variables otherwise.
VAR=$1; shift
ROUNDS=2; VALS=()
for a in "$@"; do case $a in x*) ROUNDS=${a#x};; *) VALS+=("$a");; esac; done
run() { env "$VAR=$1" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --soak-seconds 0 --allow-knobs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); e = d['extra']
print('step %.3f ms  lde %.3f  commit %.3f  fri %.3f' % (d['ms_per_step'], e['lde_ms'], e['commit_ms'], e['fri_commit']['ms']))"; }
for i in $(seq $ROUNDS); do for v in "${VALS[@]}"; do echo "$VAR=$v: $(run $v)"; done; done
