#!/bin/bash
run() { env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read())['extra']; print(round(d['commit_ms'],3), round(d['fri_commit']['ms'],3))"; }
for i in 1 2; do for t in 6 7 8 9; do echo "tail_log=$t: $(run HODOR_MERKLE_TAIL_LOG=$t)"; done; done
