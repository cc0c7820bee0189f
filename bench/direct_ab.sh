#!/bin/bash
# Same-box comparison of the 4-step step at world 1 (2^24 points): no exchange at all, the direct transport (the producing
# pass stores into the receive buffer), the copy-engine variant (on ONE device its copies are blit kernels, not SDMA), RCCL's all-to-all forced (torch.distributed and the library's own, 4 chunks,
# pipelined across steps), and the plain single-device transform.  usage: bash bench/direct_ab.sh [rounds]
R=${1:-2}
COMMON="--no-cpu-baseline --no-extra --soak-seconds 0 --steps 60 --warmup 20"
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms (strict %.3f)' % (d['ms_per_step'], d.get('ms_per_step_strict', d['ms_per_step'])))"; }
tr() { python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 1000)) bench.py --gpus 1 "$@" 2>/dev/null | grep '^{' | ms; }
for i in $(seq $R); do
  echo "single-device transform      : $(python bench.py $COMMON 2>/dev/null | ms)"
  echo "4-step, no exchange          : $(python bench.py --mode sixstep $COMMON 2>/dev/null | ms)"
  echo "4-step, direct transport     : $(python bench.py --mode sixstep --exchange direct $COMMON 2>/dev/null | ms)"
  echo "  ... coarse-grained receive  : $(python bench.py --mode sixstep --exchange direct --recv-coarse $COMMON 2>/dev/null | ms)"
  echo "4-step, copy engines, 4 chunks: $(tr --mode sixstep --force-collectives --exchange copy --exchange-chunks 4 $COMMON)"
  echo "4-step, RCCL (torch), forced : $(tr --mode sixstep --force-collectives $COMMON)"
  echo "4-step, RCCL (native), forced: $(tr --mode sixstep --force-collectives --exchange native $COMMON)"
done
