# Phase ablation of k_ntt_pass (profiling only; results are wrong by construction).  Uses the
# -DHODOR_ABLATE build (make -C hodor_amd/csrc ablate -> hodor_amd/libhodor_gpu_ablate.so); the shipped
# library contains none of these switches.
# HODOR_DBG bits: 1 skip butterflies, 2 skip inter-pass twiddles, 4 skip global loads, 8 skip global stores, 32 no
# reduction / pack / unpack of the intermediates (upper bound of what a lazy 48-byte record format could save)
export HODOR_LIB=$PWD/hodor_amd/libhodor_gpu_ablate.so
run() { echo "== $*"; env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra --allow-knobs --skip-checks 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],3))"; }
run HODOR_DBG=0
run HODOR_DBG=1
run HODOR_DBG=2
run HODOR_DBG=3
run HODOR_DBG=4
run HODOR_DBG=8
run HODOR_DBG=12
run HODOR_DBG=15
run HODOR_DBG=14
run HODOR_DBG=32
run HODOR_DBG=0
run HODOR_DBG=32
