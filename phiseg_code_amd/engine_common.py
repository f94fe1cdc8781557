"""Shared by the engine modules (engine.py: parameter store, plan construction, capture and replay; engine_forward.py: lowering of
the forward operators; engine_backward.py: the reverse-mode backward pass): dtype codes, schedule constants, device buffers."""
import ctypes
import os

import numpy as np
import torch

from phiseg_code_amd import graph as G
from phiseg_code_amd import runtime as rt
from phiseg_code_amd.tfwrapper import normalisation as tfnorm

__all__ = ['F32', 'BF16', 'U8', '_TORCH_DT', '_NP_DT', '_ESIZE', '_LIK_SIDE_MAXLVL', '_WGRAD_DEFER_BLOCKS', '_NREP', '_NREP_MINP', '_STAMPS', '_DETERMINISTIC', '_BN_SMALL', '_BN_SMALL_F32', '_BN_WIDE', '_BN_WIDE_MAXLINES', '_SKIP_HEAD_A', '_POOL_FUSE', '_xf_enabled', '_upconv_min_h', 'UpBuf', '_fgn_mode', '_dual_enabled', '_f32_mfma_enabled', '_onepass_enabled', '_noop', '_device', 'live_variables', 'device_sync', 'Buf', 'DualBuf', 'HeadGrad', 'SliceGrad', 'XfBuf']

F32, BF16, U8 = rt.F32, rt.BF16, 2
_TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16, U8: torch.uint8}
_NP_DT = {F32: np.float32, U8: np.uint8}
_ESIZE = {F32: 4, BF16: 2, U8: 1}
# two lanes: likelihood chains of levels <= this go to the prior's lane; the coarsest chain stays on lane 0, in front of the top-down
# path that starts with it: lane 0 then enters the likelihood without waiting for lane 1 (with all five chains there it started only
# when the LAST of them was done, whatever the order: 12.38 vs 11.95 ms)
_LIK_SIDE_MAXLVL = 3
_WGRAD_DEFER_BLOCKS = 96      # pixel-tile split target of a deferred layer (measured 16 .. 192: fewer slices are long tail blocks)
_NREP = 4                     # accumulator replicas of the norm backward reduction (re-measured with the LDS-shared prologues; 8 before)
_NREP_MINP = 4096
_STAMPS = os.environ.get("PHX_STAMPS", "0") == "1"           # dev: per-operator device time stamps (tools/lane_timeline.py)
_DETERMINISTIC = os.environ.get("PHX_DETERMINISTIC", "0") not in ("0", "")   # fixed summation orders everywhere (libphx reads the same variable)
_BN_SMALL = 1024              # one-launch batch norm up to this many pixels
_BN_SMALL_F32 = True          # ... with an fp32 pre-normalisation tensor (section 4 of DESIGN.md, Round 5 additions)
_BN_WIDE_MAXLINES = 2048      # ... up to this many (pixel, slice) rows per block: the 2 x 2 level (measured: 4096 = the 4 x 4 level too is 0.2 % slower, 0 = off 0.7 % slower)
_SKIP_HEAD_A = True           # the activation whose only reader is a 1x1 head is never written in a training plan (time-neutral same box, - 0.4 GB/step)
_POOL_FUSE = True             # averagepool2D of a conv unit's output is written by that unit's apply pass (phx_norm_apply_pool: 10 launches, 0.3 GB; + 0.1 % same box)
_BN_WIDE = 2                  # phx_bn_wide_fwd / _bwd take the split-K slices of the forward convolution (1) and of the data gradient too (2)


def _fgn_mode():
    # conv + bias + group norm + activation in one launch on maps <= 16 x 16 (phx_conv3x3_mfma_bf16_fgn).  1: group norm (16-channel
    # groups); 2: instance norm too (per-channel statistics: every lane adds to the LDS table -- measured 11.24 vs 11.15 ms, so not
    # by default); 0: off.  Group norm, phiseg_7_5 B = 64: 11.28 vs 11.28 - 11.30 ms with 57 launches fewer.
    return int(os.environ.get("PHX_FGN", "1"))


def _upconv_min_h():
    """bilinear_upsample2D -> conv2D 3x3 -> batch norm edges of a training plan run in the phase form (no up-sampled tensor: upconv.py,
    csrc/upconv.hip) when the LOW-resolution map is at least this high and wide and Cin >= 4 Cout; 0 = never.  PHX_UPCONV overrides (dev A/B)."""
    return int(os.environ.get("PHX_UPCONV", "64"))


def _xf_enabled():
    # conv2d -> batch_norm -> relu -> conv2d edges of the 32-channel 128 x 128 level: the apply pass and the activation tensor are
    # never made, the readers (k_conv3x3_c32, k_conv3x3_wgrad_dma<32, *>) re-form the activation in their staged patches
    # (phx_conv3x3_mfma_bf16_xf, phx_conv3x3_wgrad_mfma_bf16_partial_xf).
    # A/B hook like PHX_DUAL (read when a plan is built; tests/test_plan_variants_gpu.py compares the plan with and without).
    return os.environ.get("PHX_XF", "1") == "1"


def _dual_enabled():
    return os.environ.get("PHX_DUAL", "1") == "1"      # concat -> conv3x3 edges without the concatenated tensor (A/B hook; read when a plan is built)


def _f32_mfma_enabled():
    # fp32 plans: the 3x3 convolutions and both gradients on the fp32 matrix instruction (csrc/conv_f32_mfma.hip) instead of the vector
    # kernels of conv_direct.hip.  A/B hook (read when a plan is built); 0 = the direct kernels everywhere.
    return os.environ.get("PHX_F32_MFMA", "1") == "1"


def _onepass_enabled():
    # batch-norm backward of the mid-size layers in ONE launch (phx_bn_bwd_onepass: (dA, y) held in registers across a grid barrier)
    # instead of phx_norm_bwd_reduce + phx_norm_bwd_apply_fused.  A/B hook, read when a plan is built.
    return os.environ.get("PHX_ONEPASS", "1") == "1"


def _noop():
    pass


def _device():
    if not torch.cuda.is_available():
        raise rt.PhxError("no GPU visible: the PHiSeg engine has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def live_variables(loss):
    """Names of the variables the scalar `loss` depends on (TF: the variables optimizer.minimize gets a gradient for)."""
    seen, live, stack = set(), set(), [loss.op]
    while stack:
        op = stack.pop()
        if op in seen:
            continue
        seen.add(op)
        if op.type == "l2_weights":            # weight decay reaches every member of the collection (never-consumed branches too)
            live.update(v.name for v in op.attrs["vars"])
        if op.type == "conv_unit":
            a = op.attrs
            for v in (a["W"], a["b"]):
                if v is not None:
                    live.add(v.name)
            for v in (a.get("norm_vars") or {}).values():
                live.add(v.name)
        if op.type == "norm_act":
            for v in (op.attrs.get("norm_vars") or {}).values():
                live.add(v.name)
        stack.extend(i.op for i in op.inputs)
    return live


def device_sync():
    """Order work enqueued through torch (parameter loads, lr / step updates, broadcasts) before plan replays, which run on
    the plans' own non-blocking HIP streams."""
    torch.cuda.synchronize()


class Buf:
    """A device buffer: torch owns the memory, libphx sees the raw pointer."""

    def __init__(self, shape, dt, zero=False, like=None):
        self.shape = tuple(int(s) for s in shape)
        self.dt = dt
        n = int(np.prod(self.shape)) if self.shape else 1
        self.n = n
        if like is not None:
            self.t = like
        else:
            self.t = (torch.zeros if zero else torch.empty)(max(n, 1), dtype=_TORCH_DT[dt], device=_device())
        self.ptr = self.t.data_ptr()
        self.shift = 0           # nearest-neighbour view: logical size = stored size << shift

    @property
    def nbytes(self):
        return self.n * _ESIZE[self.dt]

    def numpy(self):
        torch.cuda.synchronize()
        a = self.t[:self.n].float().cpu().numpy() if self.dt != U8 else self.t[:self.n].cpu().numpy()
        return a.reshape(self.shape)


class DualBuf:
    """The value of tf.concat([a, b], axis=3) whose only reader is a 3x3 convolution: never materialised -- the convolution reads
    the two tensors in place (struct Dual in csrc/conv_mfma.hip), its data gradient writes their two gradients directly."""

    def __init__(self, a, b):
        self.a, self.b = a, b
        self.shape = tuple(a.shape[:-1]) + (a.shape[-1] + b.shape[-1],)
        self.dt, self.ptr, self.n, self.shift = a.dt, a.ptr, a.n + b.n, 0
        self.k1 = a.shape[-1]


class HeadGrad:
    """Placeholder for the gradient of a = act(norm(y)) whose only reader is a 1x1 head: dA = dy_head w_head^T is never materialised,
    the producer's norm backward launches form it on the fly (phx_norm_bwd_reduce_head / phx_norm_bwd_apply_fused_head)."""

    def __init__(self, like, dy, w_ptr, nout):
        self.shape, self.dt, self.n = like.shape, like.dt, like.n
        self.dy, self.w_ptr, self.nout = dy, w_ptr, nout


class SliceGrad:
    """Placeholder for the gradient of a = act(bn(y)) of a 2 x 2 / 4 x 4 layer whose only reader is a 3x3 convolution: that
    convolution's split-K data gradient skips its finishing pass and the producer's one-launch batch-norm backward sums the nz fp32
    slices ws[z][P][C] itself (phx_bn_wide_bwd) -- rounded to bf16 after the sum, as the finishing pass would have."""

    def __init__(self, like, ws, nz):
        self.shape, self.dt, self.n = like.shape, like.dt, like.n
        self.ws, self.nz = ws, nz


class XfBuf:
    """The value of a = relu(bn(y)) whose only readers are 3x3 convolutions on large maps: never materialised -- the readers'
    forward (phx_conv3x3_mfma_bf16_xf) and filter-gradient (phx_conv3x3_wgrad_*_xf) launches re-form it from the pre-normalisation
    tensor y and the layer's scale / shift in their loaders.  `ptr` is deliberately absent: nothing may read it as a tensor."""

    def __init__(self, like, y, scale, shift):
        self.shape, self.dt, self.n = like.shape, like.dt, like.n
        self.y, self.scale, self.shift = y, scale, shift


class UpBuf:
    """The value of bilinear_upsample2D(src) whose only reader is a 3x3 convolution that runs in the phase form: never materialised.
    `shape` is the hi-res shape; `ptr` is deliberately absent."""

    def __init__(self, src, shape):
        self.src, self.shape, self.dt = src, tuple(int(v) for v in shape), src.dt
        self.n = int(np.prod(self.shape))

    @property
    def nbytes(self):
        return self.n * _ESIZE[self.dt]
