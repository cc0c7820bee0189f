for R in 8 9 10; do for T in 10 11; do HODOR_MAX_LOG_R=$R HODOR_TILE_LOG=$T python bench/plan_sweep.py 16,18,20,22,24,25,26 22x8,20x16,23x8 2>/dev/null; done; done
