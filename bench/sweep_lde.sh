run() { echo "== $*"; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ntt ms_per_step', round(d['ms_per_step'],3), 'lde_ms', round(d['extra']['lde_ms'],3), 'commit_ms', round(d['extra']['commit_ms'],3))"; }
run HODOR_MAX_LOG_R=8
run HODOR_MAX_LOG_R=9
run HODOR_MAX_LOG_R=9 HODOR_TILE_LOG=10
run HODOR_MAX_LOG_R=7
run HODOR_MAX_LOG_R=7 HODOR_TILE_LOG=10
