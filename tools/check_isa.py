#!/usr/bin/env python3
"""Guards on the resident FFT kernel's code object (run on CPU; tests/test_abi_cpu.py calls it).

The kernel (opencl_fft_amd/csrc/fft_resident.hip) does two things hipcc cannot check:
  * it manages the accumulation register file by hand (literal a[N] in inline asm), so the compiler
    must not place anything of its own there: no v_accvgpr_* outside the asm blocks, no scratch;
  * it issues global loads from inline asm (hipcc neither counts nor waits for them) into AGPRs and
    into v[224:255], which amdgpu_num_vgpr(224) keeps out of the register allocator's hands: no
    compiler-generated instruction may name those registers (with compiler-allocated destinations
    hipcc was seen copying them ahead of the kernel's own s_waitcnt, i.e. before the data had landed).

usage: check_isa.py file.s   (hipcc -save-temps of fft_resident.hip)
"""
import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
AREG = re.compile(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]")
RESERVED_FIRST = 224   # v[224:255]: landing registers of asm-issued loads (kernel built with amdgpu_num_vgpr(224))


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def audit_reserved(body):
    """no instruction outside the kernel's own asm statements may name a reserved VGPR"""
    problems = []
    inasm = False
    for ln in body.split("\n"):
        s = ln.strip()
        if "ASMSTART" in s:
            inasm = True
            continue
        if "ASMEND" in s:
            inasm = False
            continue
        if inasm or not s or s.startswith((";", ".")):
            continue
        code = s.split(";")[0]
        hit = [r for r in regs_of(code) if r >= RESERVED_FIRST]
        if hit:
            problems.append("compiler instruction touches reserved v%s: %s" % (sorted(hit), code.strip()))
        # on gfx90a+ the allocator may name an AGPR directly as an operand (AV register class: ds_*, buffer_*,
        # v_mov ...) without any v_accvgpr_* instruction: the whole accumulation file is the kernel's own
        if AREG.search(code.split(None, 1)[1] if " " in code.strip() or "\t" in code.strip() else ""):
            problems.append("compiler instruction names an AGPR: %s" % code.strip())
    return problems


def audit_loop_loads(body):
    """the explicit s_waitcnt vmcnt(N) scheme counts the asm-issued loads only: a compiler-issued global / buffer
    LOAD anywhere after the table prologue would shift every count.  Compiler-issued stores only make a wait
    stronger and are allowed (the slot's write-through store, the last block's burst)."""
    problems = []
    inasm = False
    seen_barrier = False
    for ln in body.split("\n"):
        s = ln.strip()
        if "ASMSTART" in s:
            inasm = True
            continue
        if "ASMEND" in s:
            inasm = False
            continue
        if inasm or not s or s.startswith((";", ".")):
            continue
        code = s.split(";")[0].strip()
        if code.startswith("s_barrier"):
            seen_barrier = True
        if seen_barrier and re.match(r"(buffer|global|flat)_load", code):
            problems.append("compiler-issued vector load behind the prologue: %s" % code)
    return problems


def metadata_of(s, name):
    """.amdhsa / amdhsa.kernels metadata of one kernel -> dict of the integer fields"""
    m = re.search(r"- \.agpr_count:.*?\.name:\s+%s\b.*?(?=\n  - \.agpr_count:|\n\.\.\.|\Z)" % re.escape(name), s, re.S)
    out = {}
    if m:
        for k, v in re.findall(r"\.(\w+):\s+(\d+)\s*$", m.group(0), re.M):
            out[k] = int(v)
    return out


def check(path):
    s = open(path).read()
    problems = []
    found = 0
    for m in re.finditer(r"^(_ZN4clfa11k_fft_res16[A-Za-z0-9_]+):", s, re.M):
        found += 1
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
        for p in audit_reserved(body) + audit_loop_loads(body):
            problems.append("%s: %s" % (name, p))
        md = metadata_of(s, name)
        if not md:
            problems.append("%s: no kernel metadata found" % name)
        else:
            for key in ("private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count"):
                if md.get(key, 0) != 0:
                    problems.append("%s: %s = %d" % (name, key, md[key]))
    if not found:
        problems.append("no k_fft_res16 kernel in %s" % path)
    return problems


def check_handover(path):
    """conv_kernels.hip: the inter-workgroup hand-overs of k_pconv_coop / k_dconv_block read the handed-over bytes with
    agent-scope loads that must compile to global_ / buffer_ loads with sc1 (never flat_: MI355X_MICROARCH.md, 'Valid
    forms', Consumer bullet), store them with sc1 stores, drain with s_waitcnt vmcnt(0) in front of the arrival add (which
    is an agent-scope RELEASE in the C++ model as well), and —
    for launches with more workgroups than CUs — carry an agent-scope acquire (buffer_inv sc1) for the last workgroup."""
    s = open(path).read()
    problems = []
    for stem, least_loads in (("k_pconv_coop", 2), ("k_dconv_block", 8)):
        names = re.findall(r"^(_ZN4clfa\d+%s[A-Za-z0-9_]*):" % stem, s, re.M)
        if not names:
            problems.append("no %s kernel in %s" % (stem, path))
        for name in names:
            body = s[s.index("\n" + name + ":"):]
            body = body[:body.index(".Lfunc_end")]
            code = [ln.split(";")[0].strip() for ln in body.split("\n")]
            sc1_loads = [c for c in code if re.match(r"(global|buffer|flat|scratch)_load\S*\s.*\bsc1\b", c)]
            if len(sc1_loads) < least_loads:
                problems.append("%s: %d agent-scope (sc1) loads, expected at least %d" % (name, len(sc1_loads), least_loads))
            for c in sc1_loads:
                if not c.startswith(("global_load", "buffer_load")):
                    problems.append("%s: agent-scope load is not a global_/buffer_ load: %s" % (name, c))
            if any(re.match(r"flat_(load|store|atomic)", c) for c in code):
                problems.append("%s: flat_ memory instruction in a hand-over kernel" % name)
            if not any(re.match(r"(global|buffer)_store\S*\s.*\bsc1\b", c) for c in code):
                problems.append("%s: no sc1 (write-through) store of the handed-over bytes" % name)
            if not any(re.match(r"(global|buffer)_atomic_add\S*\s.*\bsc0\b", c) or re.match(r"(global|buffer)_atomic_add", c) for c in code):
                problems.append("%s: no arrival counter add" % name)
            # the arrival add carries an agent-scope release (conv_kernels.hip, handover_arrive): a write-back and its wait
            # directly in front of every counter add
            adds = [i for i, c in enumerate(code) if re.match(r"(global|buffer)_atomic_add", c)]
            for i in adds:
                before = [c for c in code[max(0, i - 6):i] if c]
                if not any(c.startswith("buffer_wbl2") and "sc1" in c for c in before) or not any(c.startswith("s_waitcnt") and "vmcnt(0)" in c for c in before):
                    problems.append("%s: counter add without an agent-scope release (buffer_wbl2 sc1 + s_waitcnt vmcnt(0)) in front of it" % name)
            if not any(c.startswith("buffer_inv") and "sc1" in c for c in code):
                problems.append("%s: no agent-scope acquire (buffer_inv sc1) for launches beyond one workgroup per CU" % name)
    return problems


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--handover":
        p = check_handover(sys.argv[2])
        for x in p:
            print(x)
        sys.exit(1 if p else 0)
    p = check(sys.argv[1])
    for x in p:
        print(x)
    sys.exit(1 if p else 0)
