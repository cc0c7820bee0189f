#!/bin/bash
# FRI commit time against codeword size with and without the fused tail kernel
for t in 0 1 0 1; do echo "HODOR_FRI_TAIL=$t"; HODOR_FRI_TAIL=$t python bench/fri_sizes.py 2>/dev/null | grep codeword; done
