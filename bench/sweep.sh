mkdir -p gpurun_out
run() { echo "== $*"; env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],4))"; }
run HODOR_NTT_THREADS=256
run HODOR_NTT_THREADS=512
run HODOR_NTT_THREADS=512 HODOR_TILE_LOG=10
run HODOR_NTT_THREADS=256 HODOR_TILE_LOG=10
run HODOR_NTT_THREADS=512 HODOR_TILE_LOG=12
run HODOR_NTT_THREADS=512 HODOR_MAX_LOG_R=6
run HODOR_NTT_THREADS=512 HODOR_MAX_LOG_R=6 HODOR_TILE_LOG=10
run HODOR_NTT_THREADS=256 HODOR_MAX_LOG_R=6 HODOR_TILE_LOG=10
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
