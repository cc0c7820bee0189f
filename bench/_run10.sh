python -m pytest tests/test_gpu_sixstep.py -x -q 2>&1 | tail -3
for c in 0 2; do HODOR_MIN_LOG_C=$c python bench/plan_sweep.py 24,25,26,27,28 22x8,23x8 2>&1 | tail -1; done
for c in 0 2; do HODOR_MIN_LOG_C=$c HODOR_MAX_LOG_R=8 python bench/plan_sweep.py 25,26,27 22x8,23x8 2>&1 | tail -1; done
HODOR_MIN_LOG_C=3 python bench/plan_sweep.py 24,25,26,27,28 22x8,23x8 2>&1 | tail -1
python -m pytest tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
