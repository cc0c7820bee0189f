#!/bin/bash
# The CPU suite against an ASAN + UBSan build of the ORACLE (oracle/hodor_oracle.c is the checker of every parity test: it
# must not lean on undefined behaviour itself).  The normal oracle is rebuilt afterwards.
# usage: bash bench/asan_oracle_suite.sh <out.log> [pytest args...]      (default: tests -m "not gpu")
OUT=${1:-/tmp/asan_oracle.log}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
[ $# -eq 0 ] && set -- "$ROOT/tests" -m "not gpu"
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c11 -shared -fPIC -pthread "$ROOT/oracle/hodor_oracle.c" \
    -o "$ROOT/oracle/libhodor_oracle.so" || exit 9
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 HODOR_TEST_ABORT_TRACE=0 \
  timeout 2400 python -m pytest "$@" -x -q -p no:cacheprovider > "$OUT" 2>&1
echo "rc=$?" >> "$OUT"
rm -f "$ROOT/oracle/libhodor_oracle.so"
make -C "$ROOT/oracle" -B libhodor_oracle.so > /dev/null
grep -n "runtime error\|ERROR: AddressSanitizer" "$OUT" | head
tail -3 "$OUT"
