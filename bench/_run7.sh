python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_sixstep.py -x -q 2>&1 | tail -3
python -m pytest tests/test_gpu_parity.py -x -q -k "fft or lde or transforms" 2>&1 | tail -3
for r in 1 2; do
for t in 0 1; do HODOR_NTT_TW_SUB=$t python bench.py --no-cpu-baseline --allow-knobs 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TWSUB', d['knobs'], round(d['ms_per_step'],4), 'lde', round(d['extra']['lde_ms'],3), 'fri', round(d['extra']['fri_commit']['ms'],3))"; done
done
