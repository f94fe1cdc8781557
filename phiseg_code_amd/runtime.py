"""ctypes binding of libphx.so (the C ABI declared in include/phx.h).

The binding is generated from the header itself, so every declared entry point is bound with the
right argument types and a missing symbol is an import-time error.  There is NO fallback: if the
HIP library is absent or a call fails, an exception is raised (the product never computes on the CPU).
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "phx.h")
DEBUG_HEADER = os.path.join(os.path.dirname(_HERE), "include", "phx_debug.h")
LIB_PATH = os.environ.get("PHX_LIB") or os.path.join(_HERE, "libphx.so")   # PHX_LIB: dev A/B builds (tools/build_variant.sh)

F32, BF16 = 0, 1
ACT_ID, ACT_RELU, ACT_SOFTPLUS = 0, 1, 2
ACT_CODES = {"identity": ACT_ID, "relu": ACT_RELU, "softplus": ACT_SOFTPLUS}
DT_SIZE = {F32: 4, BF16: 2}


class PhxError(RuntimeError):
    pass


def _ctype_of(decl):
    d = decl.strip()
    if "*" in d:
        return ctypes.c_void_p
    base = d.replace("const", "").split()
    base = base[0] if base else ""
    return {"int": ctypes.c_int, "float": ctypes.c_float, "size_t": ctypes.c_size_t,
            "uint64_t": ctypes.c_uint64, "int32_t": ctypes.c_int32}[base]


def parse_header(path=HEADER):
    """-> {name: [ctypes argtypes]} for every `int phx_*(...)` prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|size_t)\s+(phx_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        if args in ("", "void"):
            protos[name] = []
        else:
            protos[name] = [_ctype_of(a) for a in args.split(",")]
    return protos


class _Lib:
    def __init__(self, path=LIB_PATH, extra_headers=()):
        if not os.path.exists(path):
            raise PhxError("%s not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(phiseg_code_amd/csrc/build.sh); there is no CPU fallback" % (os.path.basename(path), path))
        self._dll = ctypes.CDLL(path)
        self.protos = parse_header()
        for h in extra_headers:
            self.protos.update(parse_header(h))
        for name, argtypes in self.protos.items():
            fn = getattr(self._dll, name)        # AttributeError if the symbol is not exported
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
            if name in ("phx_abi_version", "phx_last_error", "phx_conv3x3_mfma_bf16_tiles", "phx_conv3x3_mfma_bf16_tiles_dual", "phx_bn_small_supported", "phx_bn_wide_supported", "phx_norm_apply_pool_supported", "phx_upconv_supported", "phx_norm_small_supported", "phx_conv3x3_mfma_stats_atomic_supported", "phx_norm_head_supported", "phx_conv3x3_mfma_ksplit", "phx_conv3x3_fgn_supported", "phx_conv3x3_mfma_f32out_supported", "phx_conv3x3_xf_supported", "phx_conv3x3_wgrad_xf_supported", "phx_conv3x3_wgrad_multi_job_bytes", "phx_conv3x3_wgrad_ws_bytes", "phx_conv3x3_wgrad_ws_bytes_dual", "phx_augment_param_bytes",
                        "phx_conv3x3_mfma_ws_bytes", "phx_validation_metrics_ws_bytes", "phx_conv2d_direct_wgrad_ordered_ws_bytes",
                        "phx_conv3x3_f32_mfma_supported", "phx_conv3x3_f32_mfma_packed_floats", "phx_conv3x3_f32_mfma_wgrad_supported",
                        "phx_conv3x3_f32_mfma_wgrad_ws_bytes", "phx_bn_bwd_onepass_supported", "phx_bn_bwd_onepass_barrier_words"):
                if name.endswith("_ws_bytes") or name.endswith("_packed_floats"):
                    fn.restype = ctypes.c_size_t
                setattr(self, name[4:], fn)
            else:
                setattr(self, name[4:], self._checked(name, fn))

    def _checked(self, name, fn):
        def call(*args):
            rc = fn(*args)
            if rc != 0:
                buf = ctypes.create_string_buffer(512)
                self._dll.phx_last_error(buf, 512)
                raise PhxError("%s failed (%d): %s" % (name, rc, buf.value.decode(errors="replace")))
        call.__name__ = name
        return call


_lib = None
_lib_dbg = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib


def debug_lib():
    """The TEST build of the library (libphx_dbg.so: the same sources with -DPHX_DEBUG_BUILD): everything include/phx.h declares plus
    the settable kernel-selection policy of include/phx_debug.h.  For the kernel tests and dev tools only -- the engine never loads it."""
    global _lib_dbg
    if _lib_dbg is None:
        _lib_dbg = _Lib(os.path.join(_HERE, "libphx_dbg.so"), (DEBUG_HEADER,))
    return _lib_dbg


def ptr_array(ptrs):
    arr = (ctypes.c_void_p * len(ptrs))(*[ctypes.c_void_p(p) if p else None for p in ptrs])
    return arr


def int_array(vals):
    return (ctypes.c_int * len(vals))(*vals)
