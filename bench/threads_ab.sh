#!/bin/bash
run() { env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
for i in 1 2; do echo "default: $(run HODOR_NTT_THREADS=0)  128: $(run HODOR_NTT_THREADS=128)  512: $(run HODOR_NTT_THREADS=512)  384: $(run HODOR_NTT_THREADS=384)"; done
