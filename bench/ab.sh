#!/bin/bash
# A/B two builds of libhodor_gpu.so on the SAME box, interleaved (box-to-box and run-to-run noise is ~3 %).
# usage: bash bench/ab.sh hodor_amd/libhodor_gpu_base.so hodor_amd/libhodor_gpu.so [rounds]
A=$1; B=$2; N=${3:-4}
run() { HODOR_LIB=$PWD/$1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --soak-seconds 0 --allow-knobs 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in $(seq $N); do echo "A $(run $A)   B $(run $B)"; done
