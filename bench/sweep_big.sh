#!/bin/bash
# plan knobs for the sizes above 2^24 (and the LDE shapes)
for r in 7 8 9; do for t in 10 11; do
  HODOR_MAX_LOG_R=$r HODOR_TILE_LOG=$t python bench/plan_sweep.py 24,25,26,27 22x8,23x8 2>/dev/null | tail -1
done; done
