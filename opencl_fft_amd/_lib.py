"""ctypes binding of libclfft_amd.so (the C ABI in include/clfft_amd.h).

The shared library is built in-tree by ``__graft_entry__.build()`` /
``make -C opencl_fft_amd/csrc``.  There is no fallback: if the library is
missing, or it finds no HIP device, the error is raised to the caller.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (CLFA_LIB_PATH: another build of the same library, for A/B runs of bench.py — tools/build_variant.sh)
LIB_PATH = os.environ.get("CLFA_LIB_PATH") or os.path.join(_HERE, "libclfft_amd.so")
if os.environ.get("CLFA_LIB_PATH"):   # a development hook: never silently (bench.py also records the path in its line)
    import sys
    print("opencl_fft_amd: CLFA_LIB_PATH is set, loading %s instead of the in-tree library" % LIB_PATH, file=sys.stderr)

# every symbol include/clfft_amd.h declares: (name, restype, argtypes)
_vp, _fp, _ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)
SYMBOLS = [
    ("clfa_device_count", C.c_int, [_ip]),
    ("clfa_device_name", C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    ("clfa_error_string", C.c_char_p, [C.c_int]),
    ("clfa_version", C.c_char_p, []),
    ("clfa_bitrev_table", C.c_int, [C.c_int, _ip]),
    ("clfa_twiddle_table", C.c_int, [C.c_int, C.c_int, _fp]),
    ("clfa_r2c_twiddle_table", C.c_int, [C.c_int, C.c_int, _fp]),
    ("clfa_cfft_create", C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int]),
    ("clfa_rfft_create", C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int]),
    ("clfa_fft_destroy", None, [_vp]),
    ("clfa_fft_get_error", C.c_int, [_vp]),
    ("clfa_fft_get_log", C.c_char_p, [_vp]),
    ("clfa_cfft_transform", C.c_int, [_vp, _vp, C.c_long]),
    ("clfa_rfft_transform", C.c_int, [_vp, _vp, _vp, C.c_long]),
    ("clfa_fft_host_alloc", C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    ("clfa_fft_host_free", C.c_int, [_vp, _vp]),
    ("clfa_fft_exec_dev", C.c_int, [_vp, _vp, C.c_long, _vp]),
    ("clfa_fft_exec_dev_oop", C.c_int, [_vp, _vp, _vp, C.c_long, _vp]),
    ("clfa_fft_device_buffers", C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    ("clfa_fft_run_buffers", C.c_int, [_vp]),
    ("clfa_fft_device_tables", C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    ("clfa_copy_to_device", C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_int]),
    ("clfa_copy_from_device", C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_int]),
    ("clfa_stream_synchronize", C.c_int, [_vp]),
    ("clfa_fft_workspace_bytes", C.c_size_t, [_vp]),
    ("clfa_fft_kernel_name", C.c_char_p, [_vp]),
    ("clfa_reorder_dev", C.c_int, [C.c_int, _vp, _vp, C.c_int, C.c_long, _vp]),
    ("clfa_pconv_create", C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_int]),
    ("clfa_pconv_destroy", None, [_vp]),
    ("clfa_pconv_get_error", C.c_int, [_vp]),
    ("clfa_pconv_nparts", C.c_int, [_vp]),
    ("clfa_pconv_wp", C.c_int, [_vp]),
    ("clfa_pconv_wp2", C.c_int, [_vp]),
    ("clfa_pconv_push_ir", C.c_int, [_vp, _vp]),
    ("clfa_pconv_push_ir_dev", C.c_int, [_vp, _vp, C.c_long, _vp]),
    ("clfa_bandwidth_probe", C.c_int, [C.c_int, C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    ("clfa_pconv_convolution", C.c_int, [_vp, _vp, _vp]),
    ("clfa_pconv_convolution_tv", C.c_int, [_vp, _vp, _vp, _vp]),
    ("clfa_pconv_process_dev", C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    ("clfa_pconv_state_bytes", C.c_size_t, [_vp]),
    ("clfa_pconv_kernel_name", C.c_char_p, [_vp]),
    ("clfa_dconv_create", C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int]),
    ("clfa_dconv_destroy", None, [_vp]),
    ("clfa_dconv_get_error", C.c_int, [_vp]),
    ("clfa_dconv_push_ir", C.c_int, [_vp, _vp]),
    ("clfa_dconv_convolution", C.c_int, [_vp, _vp, _vp]),
    ("clfa_dconv_convolution_tv", C.c_int, [_vp, _vp, _vp, _vp]),
    ("clfa_dconv_process_dev", C.c_int, [_vp, _vp, _vp, _vp, _vp]),
]

_LIB = None


def _preload_torch_hip():
    """One HIP runtime per process.  The PyTorch-ROCm wheel ships its own
    libamdhip64.so.7 (+ its own HSA runtime) under torch/lib with the same
    soname as /opt/rocm's; the dynamic loader keeps whichever is loaded first,
    and a process that mixes the two loses its GPUs.  Callers use torch for
    device memory and streams, so when torch is installed its runtime is loaded
    first and libclfft_amd.so binds to it (found by spec, torch is not imported)."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def lib():
    """Load libclfft_amd.so once; raises OSError if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C opencl_fft_amd/csrc`" % LIB_PATH)
        _preload_torch_hip()
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            f = getattr(L, name)          # AttributeError if the ABI is incomplete
            f.restype = res
            f.argtypes = args
        _LIB = L
    return _LIB


class ClError(RuntimeError):
    """An OpenCL-numbered status returned by the library (0 is success)."""

    def __init__(self, code, where=""):
        self.code = code
        msg = lib().clfa_error_string(code).decode()
        super().__init__("%s: %s (%d)" % (where, msg, code) if where else "%s (%d)" % (msg, code))


def check(code, where=""):
    if code != 0:
        raise ClError(code, where)
    return code
