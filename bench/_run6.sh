for r in 1 2; do
for t in 0 1 2 4 8; do HODOR_NTT_TILES=$t python bench.py --no-cpu-baseline --no-extra --allow-knobs 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TILES', d['knobs'], round(d['ms_per_step'],4), d['checks'])"; done
done
python -m pytest tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
