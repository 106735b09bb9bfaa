"""CPU-side checks of the drop-in boundary: libclfft_amd.so loads, exports every
symbol include/clfft_amd.h declares, and its host-only entry points (tables,
error strings) match the oracle bit for bit.  No compute calls: there is no GPU
here, and the library has no CPU fallback (constructors must say so)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import opencl_fft_amd as fa
from opencl_fft_amd import _lib
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "clfft_amd.h")).read()
    return sorted(set(re.findall(r"CLFA_API[^;(]*?\b(clfa_\w+)\s*\(", src)))


def test_header_symbols_all_exported():
    names = _declared()
    assert len(names) >= 35
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), "libclfft_amd.so does not export %s" % n
    bound = {s[0] for s in _lib.SYMBOLS}
    assert bound == set(names), (bound ^ set(names))


def test_no_oracle_or_cpu_fallback_linked():
    """the product library must not contain the oracle"""
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    assert "orc_" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "opencl_fft_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), "%s mentions the oracle" % f


@pytest.mark.parametrize("n", [2, 16, 1024, 65536])
def test_tables_match_oracle_bit_exact(n):
    assert np.array_equal(fa.bitrev_table(n), oracle.bitrev_table(n))
    for fwd in (True, False):
        assert np.array_equal(fa.twiddle_table(n, fwd).view(np.uint32), oracle.twiddle_table(n, fwd).view(np.uint32))
        assert np.array_equal(fa.r2c_twiddle_table(n, fwd).view(np.uint32),
                              oracle.r2c_twiddle_table(n, fwd).view(np.uint32))


def test_error_strings_follow_cl_numbering():
    # cl_fft.cpp:298-395
    assert fa.cl_error_string(0) == "Success!"
    assert fa.cl_error_string(-1) == "Device not found."
    assert fa.cl_error_string(-5) == "Out of resources"
    assert fa.cl_error_string(-30) == "Invalid value"
    assert fa.cl_error_string(-54) == "Invalid work group size"
    assert fa.cl_error_string(-62) == "Invalid mip-map level"
    assert fa.cl_error_string(12345) == "Unknown error"


def test_fails_loudly_without_a_device():
    if fa.device_count() > 0:
        pytest.skip("a HIP device is present")
    plan = fa.Clcfft(0, 1024, True)
    assert plan.get_error() == -1 and fa.cl_error_string(plan.get_error()) == "Device not found."
    x = np.ones(1024, np.complex64)
    assert plan.transform(x) == -1
    assert np.all(x == 1)                       # untouched: nothing computed on the CPU
    assert fa.Clrfft(0, 1024, True).get_error() == -1
    msgs = []
    pc = fa.Clpconv(0, 4096, 1024, errs=lambda s, d: msgs.append(s))
    assert pc.get_cl_err() == -1 and msgs == ["Device not found."]
    assert fa.Cldconv(0, 64, 8, uData=object()).get_cl_err() == -1


def test_engine_emulation_on_cpu(tmp_path):
    """the lane-level pass code of fft_device.hpp, run lane by lane on the host"""
    exe = str(tmp_path / "emulate_engine")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "emulate_engine.cpp"),
                           "-o", exe])
    out = subprocess.check_output([exe]).decode()
    assert out.strip().endswith("OK"), out


def _compile_to_asm(tmp_path, name):
    src = os.path.join(ROOT, "opencl_fft_amd", "csrc", name + ".hip")
    flags = None
    for ln in open(os.path.join(ROOT, "opencl_fft_amd", "csrc", "Makefile")):
        if ln.startswith("CXXFLAGS"):
            flags = ln.split("=", 1)[1].replace("$(ARCH)", "gfx950").split()
    assert flags, "CXXFLAGS not found in the Makefile"
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this machine: the code-object audit needs the gfx950 cross-compiler")
    p = subprocess.run([hipcc] + flags + ["-save-temps=obj", "-c", src, "-o", str(tmp_path / (name + ".o"))],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert p.returncode == 0, p.stdout.decode()[-4000:]
    asm = [f for f in os.listdir(tmp_path) if f.endswith("gfx950.s")]
    assert len(asm) == 1, asm
    return str(tmp_path / asm[0])


def _check_isa():
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_isa", os.path.join(ROOT, "tools", "check_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_handover_kernels_code_object_audit(tmp_path):
    """conv_kernels.hip: the hand-overs of k_pconv_coop / k_dconv_block on the compiled ISA — sc1 stores, agent-scope loads
    as global_ / buffer_ loads with sc1 (never flat_), an arrival add, and the acquire for launches beyond one workgroup
    per CU (tools/check_isa.py --handover)"""
    problems = _check_isa().check_handover(_compile_to_asm(tmp_path, "conv_kernels"))
    assert not problems, "\n".join(problems[:10])


def test_tiny_kernel_code_object_audit(tmp_path):
    """fft_kernels.hip: k_fft_tiny<2> (n = 4) exchanges the halves of a transform between its two lanes through the DPP crossbar —
    on the compiled ISA: quad_perm:[1,0,3,2] operands, 16-byte loads and stores, no LDS instruction, no scratch; and the
    pass-to-pass twiddles of the column kernels above 65536 points come from LDS (no global table lookups: their only
    vector-memory instructions are the 16 + 16 that move a lane's data ... 32 + 32 for the two-run forms)"""
    import re
    text = open(_compile_to_asm(tmp_path, "fft_kernels")).read()
    bodies = {}
    for m in re.finditer(r"^(_ZN4clfa\w+):.*?\n(.*?)s_endpgm", text, re.S | re.M):
        bodies[m.group(1)] = m.group(2)
    tiny4 = [k for k in bodies if "10k_fft_tinyILi2E" in k]
    assert len(tiny4) >= 3, sorted(bodies)[:5]
    for k in tiny4:
        b = bodies[k]
        assert "quad_perm:[1,0,3,2]" in b, k
        assert "global_load_dwordx4" in b and "global_store_dwordx4" in b, k
        assert not re.search(r"\bds_\w+", b) and "scratch_" not in b, k
    for k in [k for k in bodies if "10k_fft_tinyILi1E" in k]:
        assert "dpp" not in bodies[k] and not re.search(r"\bds_\w+", bodies[k]), k
    cols = [k for k in bodies if "14k_big2_cols_2x" in k or "11k_big2_colsI" in k]
    assert len(cols) >= 6, len(cols)
    for k in cols:
        loads = len(re.findall(r"\b(global|buffer|flat)_load_\w+", bodies[k]))
        per_lane = 32 if "cols_2x" in k else 16
        assert loads <= per_lane + 12, (k, loads)   # data + the tables' copy into LDS at the start


def test_resident_kernel_code_object_audit(tmp_path):
    """fft_resident.hip manages the accumulation registers and v[224:255] by hand; tools/check_isa.py
    verifies on the freshly compiled ISA that hipcc put nothing of its own there and uses no scratch"""
    import importlib.util
    src = os.path.join(ROOT, "opencl_fft_amd", "csrc", "fft_resident.hip")
    flags = None
    for ln in open(os.path.join(ROOT, "opencl_fft_amd", "csrc", "Makefile")):
        if ln.startswith("CXXFLAGS"):
            flags = ln.split("=", 1)[1].replace("$(ARCH)", "gfx950").split()
    assert flags, "CXXFLAGS not found in the Makefile"
    obj = str(tmp_path / "fft_resident.o")
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this machine: the code-object audit needs the gfx950 cross-compiler")
    p = subprocess.run([hipcc] + flags + ["-save-temps=obj", "-c", src, "-o", obj], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT)
    assert p.returncode == 0, p.stdout.decode()[-4000:]
    asm = [f for f in os.listdir(tmp_path) if f.endswith("gfx950.s")]
    assert len(asm) == 1, asm
    spec = importlib.util.spec_from_file_location("check_isa", os.path.join(ROOT, "tools", "check_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    problems = mod.check(str(tmp_path / asm[0]))
    assert not problems, "\n".join(problems[:10])
