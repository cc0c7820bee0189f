#!/bin/bash
# The GPU suite (or a selection of it) against the ASAN + UBSan build of the library's HOST side (make -C hodor_amd/csrc asan):
# a heap overflow / use-after-free / UB in the C ABI's host code is reported where it happens, not where glibc or a
# runtime thread later trips over the damage.
# usage: bash bench/asan_suite.sh <out.log> [pytest args...]
OUT=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -B -C "$ROOT/hodor_amd/csrc" asan > /dev/null   # -B: a directory named asan/ exists, make would call the target up to date || exit 9
ASAN=$(hipcc -print-file-name=libclang_rt.asan-x86_64.so)
mkdir -p "$(dirname "$OUT")"
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1:protect_shadow_gap=0:log_path="$OUT.asan" \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1:log_path="$OUT.ubsan" \
  HODOR_LIB="$ROOT/hodor_amd/libhodor_gpu_asan.so" HODOR_TEST_ABORT_TRACE=0 \
  timeout 2400 python -m pytest "$@" -x -q -p no:cacheprovider > "$OUT" 2>&1
echo "rc=$?" >> "$OUT"
tail -5 "$OUT"
ls -la "$OUT".asan* "$OUT".ubsan* 2>/dev/null
