#!/bin/bash
# Same-box interleaved A/B of several builds of the library (HODOR_LIB) on the whole bench line.
# usage: bash bench/lib_ab.sh [xROUNDS] a.so b.so ...
ROUNDS=2; LIBS=()
for a in "$@"; do case $a in x*) ROUNDS=${a#x};; *) LIBS+=("$a");; esac; done
run() { HODOR_LIB=$PWD/$1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --soak-seconds 0 --allow-knobs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); e = d['extra']
print('step %.3f ms  lde %.3f  commit %.3f  fri %.3f' % (d['ms_per_step'], e['lde_ms'], e['commit_ms'], e['fri_commit']['ms']))"; }
for i in $(seq $ROUNDS); do for l in "${LIBS[@]}"; do echo "$l: $(run $l)"; done; done
