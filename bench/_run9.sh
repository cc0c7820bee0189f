for r in 7 8 9; do HODOR_MAX_LOG_R=$r python bench/plan_sweep.py 23,24,25,26,27 22x8,23x8 2>&1 | tail -1; done
for r in 8 9; do HODOR_MAX_LOG_R=$r HODOR_TILE_LOG=11 python bench/plan_sweep.py 24,25,26,27 22x8 2>&1 | tail -1; done
