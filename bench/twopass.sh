#!/bin/bash
# The two-pass plan 2^24 = 2^12 x 2^12 for real (experiment build: -DHODOR_TWOPASS raises the workgroup to 1024
# threads and the radix cap to 2^12; see the comment in hodor_amd/csrc/ntt.hip): one 4096-point tile per CU.
#   hipcc ... -DHODOR_TWOPASS -c ntt.hip abi_host.hip ; link as hodor_amd/libhodor_gpu_2pass.so ; bash bench/twopass.sh
export HODOR_LIB=$PWD/hodor_amd/libhodor_gpu_2pass.so
run() { echo "== $*"; env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra --allow-knobs 2>&1 | python -c "
import sys, json
t = sys.stdin.read()
try:
    d = json.loads(t.strip().splitlines()[-1]); print('ms_per_step %.3f' % d['ms_per_step'], d['checks'])
except Exception: print(t[-600:])"; }
run HODOR_MAX_LOG_R=9
run HODOR_MAX_LOG_R=12 HODOR_TILE_LOG=12 HODOR_MIN_LOG_C=0
run HODOR_MAX_LOG_R=9
run HODOR_MAX_LOG_R=12 HODOR_TILE_LOG=12 HODOR_MIN_LOG_C=0
