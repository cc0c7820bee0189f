#!/usr/bin/env python3
"""Static ISA histogram of one kernel of a gfx950 assembly listing (hipcc --cuda-device-only -S).

usage: python bench/isa_hist.py [file.hip|file.s] [kernel-name-substring] [top]
A .hip argument is compiled first (same flags as hodor_amd/csrc/Makefile).  The histogram is STATIC (every
instruction of the kernel's text counted once); bench/profile.sh's SQ_INSTS_VALU pass gives the dynamic count.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

HALF_RATE = ("v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_lshrrev_b64", "v_lshlrev_b64", "v_lshl_add_u64",
             "v_alignbit_b32", "v_add3_u32", "v_addc_co_u32", "v_subb_co_u32", "v_perm_b32", "v_bfi_b32",
             "v_lshl_or_b32", "v_and_or_b32", "v_xad_u32", "v_mad_u32_u24", "v_lshl_add_u32", "v_add_lshl_u32",
             "v_or3_b32", "v_bfe_u32")


def listing(path, extra=()):
    if path.endswith(".s"):
        return open(path).read()
    out = tempfile.mktemp(suffix=".s")
    root = os.path.dirname(os.path.abspath(path))
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function",
                           "-Wno-unused-command-line-argument", "--cuda-device-only", "-S", path, "-o", out, *extra],
                          cwd=root)
    text = open(out).read()
    os.unlink(out)
    return text


def kernel_body(text, name):
    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(name) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.section") or ".Lfunc_end" in lines[i])
    return lines[start].split(":")[0], lines[start + 1:end]


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "hodor_amd", "csrc", "ntt.hip")
    name = sys.argv[2] if len(sys.argv) > 2 else "k_ntt_passILi0E"
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    extra = sys.argv[4:]
    text = listing(path, extra)
    sym, body = kernel_body(text, name)
    hist = collections.Counter()
    for l in body:
        l = l.strip()
        if not l or l.startswith((";", ".", "s_nop")) and not l.startswith("s_nop"):
            continue
        if l.endswith(":") or l.startswith(";") or l.startswith("."):
            continue
        hist[l.split()[0]] += 1
    valu = sum(c for k, c in hist.items() if k.startswith("v_") and not k.startswith("v_cmp") or k.startswith("v_cmp"))
    half = sum(c for k, c in hist.items() if k.split("_e64")[0] in HALF_RATE)
    m = re.search(re.escape(sym) + r".*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", text, re.S)
    spill = re.search(re.escape(sym) + r".*?\.vgpr_spill_count:\s+(\d+)", text, re.S)
    print(f"{sym}")
    print(f"  sgpr {m.group(1)} vgpr {m.group(2)} vgpr_spills {spill.group(1) if spill else '?'}")
    print(f"  static: VALU {valu}  half-rate forms {half}  issue slots (full=1, half=2) {valu + half}"
          f"  s_nop {hist.get('s_nop', 0)}  ds {sum(c for k, c in hist.items() if k.startswith('ds_'))}"
          f"  readlane/writelane {hist.get('v_readlane_b32', 0) + hist.get('v_writelane_b32', 0)}")
    for k, c in hist.most_common(top):
        print(f"  {c:6d}  {k}")


if __name__ == "__main__":
    main()
