#!/bin/bash
# same-box interleaved A/B of the FRI commit and LDE+commit timings for two builds of the library
A=$1; B=$2; N=${3:-3}
run() { HODOR_LIB=$PWD/$1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read())['extra']; print(round(d['commit_ms'],3), round(d['fri_commit']['ms'],3))"; }
for i in $(seq $N); do echo "A $(run $A)   B $(run $B)"; done
