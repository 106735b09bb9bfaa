#!/usr/bin/env python3
"""Guards on the resident FFT kernel's code object (run on CPU; tests/test_abi_cpu.py calls it).

The kernel manages the accumulation register file by hand (literal a[N] numbers in inline asm), so
the compiler must not place anything of its own there: no v_accvgpr_* outside the asm blocks, no
scratch memory, and the arch VGPR count must stay within 256.

usage: check_isa.py file.s   (hipcc -save-temps of fft_resident.hip)
"""
import re
import sys


def check(path):
    s = open(path).read()
    problems = []
    for m in re.finditer(r"^(_ZN4clfa11k_fft_res16[A-Za-z0-9_]+):", s, re.M):
        name = m.group(1)
        body = s[m.end():]
        body = body[:body.index(".Lfunc_end")]
        inasm = False
        stray = 0
        for ln in body.split("\n"):
            if "ASMSTART" in ln:
                inasm = True
            elif "ASMEND" in ln:
                inasm = False
            elif "accvgpr" in ln and not inasm:
                stray += 1
        if stray:
            problems.append("%s: %d compiler-generated AGPR moves" % (name, stray))
        if re.search(r"\bscratch_(load|store)", body):
            problems.append("%s: scratch memory accesses" % name)
    return problems


if __name__ == "__main__":
    p = check(sys.argv[1])
    for x in p:
        print(x)
    sys.exit(1 if p else 0)
