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


class Conv3x3Desc(ctypes.Structure):
    """phx_conv3x3_desc (include/phx.h): the one descriptor of the bf16 MFMA forward / data-gradient launch."""
    _fields_ = ([(n, ctypes.c_void_p) for n in ("x", "x2", "xscale", "xshift", "wpk", "y", "y2", "y_f32", "bias", "oscale", "stats", "workspace")]
                + [("workspace_bytes", ctypes.c_size_t)]
                + [(n, ctypes.c_void_p) for n in ("a_out", "gamma", "beta", "mean_out", "rstd_out", "scale_out", "shift_out")]
                + [(n, ctypes.c_int) for n in ("K1", "N1", "act", "stats_mode", "sum_slices", "gn_groups")]
                + [("gn_eps", ctypes.c_float)] + [(n, ctypes.c_int) for n in ("B", "H", "W", "K", "N")] + [("reserved", ctypes.c_int * 4)])


class Conv3x3Plan(ctypes.Structure):
    """phx_conv3x3_plan (include/phx.h)."""
    _fields_ = [(n, ctypes.c_int) for n in ("tiles", "tiles_dual", "ksplit", "stats_atomic_ok", "f32out_ok", "xf_ok", "fgn_block", "reserved")] + \
               [("ws_bytes", ctypes.c_size_t)]


def _p(v):
    """pointer argument of the host shims: int address, ctypes pointer or None"""
    if v is None:
        return None
    return v if isinstance(v, int) else ctypes.cast(v, ctypes.c_void_p).value


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
            if name in ("phx_abi_version", "phx_last_error", "phx_bn_small_supported", "phx_bn_wide_supported", "phx_norm_apply_pool_supported", "phx_upconv_supported", "phx_norm_small_supported", "phx_norm_head_supported", "phx_conv3x3_wgrad_xf_supported", "phx_conv3x3_wgrad_multi_job_bytes", "phx_conv3x3_wgrad_ws_bytes", "phx_conv3x3_wgrad_ws_bytes_dual", "phx_augment_param_bytes",
                        "phx_conv3x3_desc_bytes", "phx_validation_metrics_ws_bytes", "phx_conv2d_direct_wgrad_ordered_ws_bytes",
                        "phx_conv3x3_f32_mfma_supported", "phx_conv3x3_f32_mfma_packed_floats", "phx_conv3x3_f32_mfma_wgrad_supported",
                        "phx_conv3x3_f32_mfma_wgrad_ws_bytes", "phx_bn_bwd_onepass_supported", "phx_bn_bwd_onepass_barrier_words"):
                if name.endswith("_ws_bytes") or name.endswith("_packed_floats"):
                    fn.restype = ctypes.c_size_t
                setattr(self, name[4:], fn)
            else:
                setattr(self, name[4:], self._checked(name, fn))

        assert self._dll.phx_conv3x3_desc_bytes() == ctypes.sizeof(Conv3x3Desc), "phx_conv3x3_desc layout mismatch (include/phx.h vs runtime.py)"
        self._conv_shims()

    # ---- host-side shims of the bf16 3x3 convolution family ---------------------------------------------------------------
    # The C ABI has ONE launch entry (phx_conv3x3_bf16, a descriptor struct) and ONE plan query (phx_conv3x3_bf16_plan) for the forward /
    # data-gradient family.  The engine, the kernel tests and the dev tools name the variants they launch; these Python callables keep
    # those names (and signatures) and fill the descriptor.  They are host conveniences, not part of the ABI.
    def conv3x3_plan(self, B, H, W, K, N, G=0):
        pl = Conv3x3Plan()
        self.conv3x3_bf16_plan(B, H, W, K, N, G, ctypes.byref(pl))
        return pl

    def _conv_shims(self):
        launch = self.conv3x3_bf16
        ptr_fields = {n for n, t in Conv3x3Desc._fields_ if t is ctypes.c_void_p}

        def mk(name, build):
            def call(*args):
                d = build(*args[:-1])
                launch(ctypes.byref(d), args[-1])
            call.__name__ = name
            setattr(self, name[4:], call)

        def desc(B, H, W, K, N, **kw):
            d = Conv3x3Desc()
            d.B, d.H, d.W, d.K, d.N = B, H, W, K, N
            for k, v in kw.items():
                setattr(d, k, _p(v) if k in ptr_fields else v)
            return d
        mk("phx_conv3x3_mfma_bf16", lambda x, wpk, y, bias, act, sp, B, H, W, K, N: desc(
            B, H, W, K, N, x=x, wpk=wpk, y=y, bias=bias, act=act, stats=sp, stats_mode=1 if _p(sp) else 0))
        mk("phx_conv3x3_mfma_bf16_ws", lambda x, wpk, y, bias, act, sp, ws, wsb, B, H, W, K, N: desc(
            B, H, W, K, N, x=x, wpk=wpk, y=y, bias=bias, act=act, stats=sp, stats_mode=1 if _p(sp) else 0, workspace=ws, workspace_bytes=wsb))
        mk("phx_conv3x3_mfma_bf16_stats_atomic", lambda x, wpk, y, bias, act, sums, B, H, W, K, N: desc(
            B, H, W, K, N, x=x, wpk=wpk, y=y, bias=bias, act=act, stats=sums, stats_mode=2))
        mk("phx_conv3x3_mfma_bf16_dual", lambda x, x2, K1, wpk, y, y2, N1, bias, oscale, act, stats, mode, ws, wsb, B, H, W, K, N: desc(
            B, H, W, K, N, x=x, x2=x2, K1=K1, wpk=wpk, y=y, y2=y2, N1=N1, bias=bias, oscale=oscale, act=act, stats=stats, stats_mode=mode,
            workspace=ws, workspace_bytes=wsb))
        mk("phx_conv3x3_mfma_bf16_f32out", lambda x, x2, K1, wpk, yf, sum_slices, ws, wsb, B, H, W, K, N: desc(
            B, H, W, K, N, x=x, x2=x2, K1=K1, wpk=wpk, y_f32=yf, sum_slices=sum_slices, workspace=ws, workspace_bytes=wsb))
        mk("phx_conv3x3_mfma_bf16_xf", lambda x, xscale, xshift, wpk, y, sp, B, H, W, K, N: desc(
            B, H, W, K, N, x=x, xscale=xscale, xshift=xshift, wpk=wpk, y=y, stats=sp, stats_mode=1 if _p(sp) else 0))
        mk("phx_conv3x3_mfma_bf16_fgn", lambda x, wpk, y, a_out, bias, gamma, beta, eps, G, act, mean, rstd, scale, shift, B, H, W, K, N: desc(
            B, H, W, K, N, x=x, wpk=wpk, y=y, a_out=a_out, bias=bias, gamma=gamma, beta=beta, gn_eps=eps, gn_groups=G, act=act, mean_out=mean,
            rstd_out=rstd, scale_out=scale, shift_out=shift))
        mk("phx_conv3x3_mfma_bf16_affine", lambda x, wpk, y, scale, shift, act, ws, wsb, B, H, W, K, N: desc(
            B, H, W, K, N, x=x, wpk=wpk, y=y, oscale=scale, bias=shift, act=act, workspace=ws, workspace_bytes=wsb))
        q = self.conv3x3_plan
        self.conv3x3_mfma_bf16_tiles = lambda B, H, W, K, N: q(B, H, W, K, N).tiles
        self.conv3x3_mfma_bf16_tiles_dual = lambda B, H, W, K, N: q(B, H, W, K, N).tiles_dual
        self.conv3x3_mfma_ws_bytes = lambda B, H, W, K, N: q(B, H, W, K, N).ws_bytes
        self.conv3x3_mfma_ksplit = lambda B, H, W, K, N: q(B, H, W, K, N).ksplit
        self.conv3x3_mfma_stats_atomic_supported = lambda B, H, W, K, N: q(B, H, W, K, N).stats_atomic_ok
        self.conv3x3_mfma_f32out_supported = lambda B, H, W, K, N: q(B, H, W, K, N).f32out_ok
        self.conv3x3_xf_supported = lambda B, H, W, K, N: q(B, H, W, K, N).xf_ok
        self.conv3x3_fgn_supported = lambda B, H, W, K, N, G: q(B, H, W, K, N, G).fgn_block

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
