#!/bin/bash
# Same-box A/B of builds of the library on the FRI commit of a 2^26 codeword, both tree formats (best of 4 after a warm-up).
# usage: bash bench/fri_ab.sh a.so b.so ...
run() { HODOR_LIB=$PWD/$1 python bench/fri_gap.py 26 $2 2>/dev/null | sort -t: -k2 -n | head -1; }
for i in 1 2 3; do for l in "$@"; do echo "$l: $(run $l) | $(run $l coset2)"; done; done
