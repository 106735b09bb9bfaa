#!/usr/bin/env python3
"""Per-basic-block instruction census of one kernel in a hipcc .s file (-save-temps).

usage: isa_blocks.py file.s <kernel-symbol-substring> [--min N]

Prints, for every label-delimited block of the kernel, the number of instructions by class
(VALU, packed VALU, LDS, global/scratch memory, SALU, waitcnt, barrier, branch) so that the dynamic
instruction count of a loop nest can be read off (block count x trip count).  The issue budget of
a CU is one VALU/LDS/VMEM instruction per SIMD per ~4 cycles, so these counts, not the flops, say
whether a "bandwidth-bound" kernel is in fact issue-bound.
"""
import re
import sys
from collections import OrderedDict


def classify(op):
    if op.startswith("v_pk_"):
        return "vpk"
    if op.startswith("v_accvgpr"):
        return "acc"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        if "load" in op:
            return "vld"
        if "store" in op:
            return "vst"
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "bar"
    if op.startswith(("s_cbranch", "s_branch")):
        return "br"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    minn = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 0
    lines = open(path).read().split("\n")
    start = None
    for i, ln in enumerate(lines):
        if ln.startswith("_Z") and key in ln and ln.rstrip().split(":")[0].endswith(ln.split(":")[0]) and ":" in ln:
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = {}
    total = {}
    for ln in lines[start + 1:]:
        s = ln.strip()
        if s.startswith(".Lfunc_end") or s.startswith("s_endpgm"):
            if s.startswith("s_endpgm"):
                blocks[cur]["salu"] = blocks[cur].get("salu", 0) + 1
            if s.startswith(".Lfunc_end"):
                break
            continue
        m = re.match(r"^(\.LBB[0-9_]+):", s)
        if m:
            cur = m.group(1)
            blocks[cur] = {}
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        c = classify(op)
        blocks[cur][c] = blocks[cur].get(c, 0) + 1
        total[c] = total.get(c, 0) + 1
    cols = ["valu", "vpk", "acc", "lds", "vld", "vst", "vmem", "salu", "wait", "bar", "br", "nop", "other"]
    print("%-14s %6s | " % ("block", "all") + " ".join("%5s" % c for c in cols))
    for name, d in blocks.items():
        n = sum(d.values())
        if n < minn:
            continue
        print("%-14s %6d | " % (name, n) + " ".join("%5d" % d.get(c, 0) for c in cols))
    print("%-14s %6d | " % ("TOTAL", sum(total.values())) + " ".join("%5d" % total.get(c, 0) for c in cols))


if __name__ == "__main__":
    main()
