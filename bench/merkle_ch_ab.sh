#!/bin/bash
# Merkle commit (2^25 leaves) and FRI commit (2^26) against the throughput chunk size and tail cut
run() { env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read())['extra']; print(round(d['commit_ms'],3), round(d['fri_commit']['ms'],3))"; }
for i in 1 2; do
  echo "ch11 tail6: $(run HODOR_LIB=$PWD/hodor_amd/libhodor_gpu.so)"
  echo "ch10 tail6: $(run HODOR_LIB=$PWD/hodor_amd/libhodor_gpu_ch10.so)"
  echo "ch10 tail5: $(run HODOR_LIB=$PWD/hodor_amd/libhodor_gpu_ch10.so HODOR_MERKLE_TAIL_LOG=5)"
  echo "ch9 tail6: $(run HODOR_LIB=$PWD/hodor_amd/libhodor_gpu_ch9.so)"
  echo "ch9 tail5: $(run HODOR_LIB=$PWD/hodor_amd/libhodor_gpu_ch9.so HODOR_MERKLE_TAIL_LOG=5)"
done
