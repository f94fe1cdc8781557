"""layers of tfwrapper/layers.py without a call site in the shipped experiments (csrc/gconv.hip) --
strided / dilated convolution, max pool, pad_to_size / crop, dropout -- kernels through the C ABI against the oracle (torch
autograd of oracle/tf1_ops.py), fp32 and bf16 storage, odd sizes, then the layer functions through the engine."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import tf1_ops as T

pytestmark = pytest.mark.gpu
F32, BF16 = 0, 1
RNG = np.random.default_rng(11)


@pytest.fixture(scope="module")
def L():
    from phiseg_code_amd import runtime as rt
    return rt.lib()


def S():
    return torch.cuda.current_stream().cuda_stream


def tdt(dt):
    return torch.float32 if dt == F32 else torch.bfloat16


def dev(a, dt=F32):
    return torch.as_tensor(np.asarray(a, dtype=np.float32)).to(tdt(dt)).cuda()


def rounded(a, dt):
    return torch.as_tensor(np.asarray(a, dtype=np.float32)).to(tdt(dt)).double()


def close(a, b, tol, what=""):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b).max() / max(1e-12, np.abs(b).max())
    assert err < tol, (what, err)


@pytest.mark.parametrize("case", [(2, 9, 7, 5, 6, 3, 3, 1, 1, 2, 2, F32), (2, 8, 8, 4, 8, 3, 3, 2, 2, 1, 1, F32), (1, 11, 6, 3, 4, 1, 1, 2, 2, 1, 1, F32),
                                  (2, 10, 10, 8, 8, 3, 3, 1, 1, 4, 4, F32), (3, 7, 9, 16, 8, 5, 3, 2, 1, 1, 2, BF16), (2, 16, 16, 32, 32, 3, 3, 1, 1, 1, 1, BF16)])
def test_general_conv_fwd_dgrad_wgrad(L, case):
    B, H, W, Ci, Co, kh, kw, sh, sw, dh, dw, dt = case
    x = RNG.standard_normal((B, H, W, Ci))
    w = RNG.standard_normal((kh, kw, Ci, Co)) / np.sqrt(kh * kw * Ci)
    b = RNG.standard_normal(Co) * 0.3
    xr = rounded(x, dt).requires_grad_(True)
    wr = torch.as_tensor(w, dtype=torch.float32).double().requires_grad_(True)
    yr = T.relu(T.conv2d_general_same(xr, wr, (sh, sw), (dh, dw)) + torch.as_tensor(b, dtype=torch.float32).double())
    ho, wo = ctypes.c_int(), ctypes.c_int()
    L.gconv2d_out_size(H, W, sh, sw, ctypes.byref(ho), ctypes.byref(wo))
    assert (ho.value, wo.value) == tuple(yr.shape[1:3])
    xd, wd, bd = dev(x, dt), dev(w), dev(b)
    y = torch.empty(B, ho.value, wo.value, Co, dtype=tdt(dt)).cuda()
    geo = (B, H, W, Ci, Co, kh, kw, sh, sw, dh, dw)
    L.gconv2d_fwd(xd.data_ptr(), dt, wd.data_ptr(), bd.data_ptr(), y.data_ptr(), dt, *geo, 1, S())
    tol = 2e-5 if dt == F32 else 8e-3
    close(y.float().cpu().numpy(), yr.detach().numpy(), tol, "fwd")
    dy = RNG.standard_normal(tuple(yr.shape))
    pre = T.conv2d_general_same(xr, wr, (sh, sw), (dh, dw))              # gradients of the un-activated convolution
    (pre * rounded(dy, dt)).sum().backward()
    dyd = dev(dy, dt)
    dx = torch.empty_like(xd)
    L.gconv2d_dgrad(dyd.data_ptr(), dt, wd.data_ptr(), dx.data_ptr(), dt, *geo, S())
    close(dx.float().cpu().numpy(), xr.grad.numpy(), tol, "dgrad")
    dwd = torch.full((kh, kw, Ci, Co), 0.25, dtype=torch.float32).cuda()
    L.gconv2d_wgrad(xd.data_ptr(), dt, dyd.data_ptr(), dt, dwd.data_ptr(), *geo, S())
    close(dwd.cpu().numpy() - 0.25, wr.grad.numpy(), 3e-5 if dt == F32 else 2e-5, "wgrad (accumulates)")


@pytest.mark.parametrize("case", [(2, 8, 8, 16, F32), (3, 7, 5, 6, F32), (2, 1, 9, 8, BF16), (2, 16, 16, 32, BF16)])
def test_maxpool_fwd_bwd(L, case):
    B, H, W, C, dt = case
    x = RNG.standard_normal((B, H, W, C))
    x[0, :2, :2, 0] = 0.0                                               # a window of ties (ReLU zeros): first element wins
    xr = rounded(x, dt).requires_grad_(True)
    yr = T.max_pool_2x2_same(xr)
    xd = dev(x, dt)
    y = torch.empty(B, (H + 1) // 2, (W + 1) // 2, C, dtype=tdt(dt)).cuda()
    L.maxpool2x2_fwd(xd.data_ptr(), dt, y.data_ptr(), B, H, W, C, S())
    close(y.float().cpu().numpy(), yr.detach().numpy(), 1e-7, "maxpool fwd")
    dy = RNG.standard_normal(tuple(yr.shape))
    (yr * rounded(dy, dt)).sum().backward()
    dyd = dev(dy, dt)
    dx = torch.full_like(xd, 7.0)
    L.maxpool2x2_bwd(xd.data_ptr(), dyd.data_ptr(), dt, dx.data_ptr(), B, H, W, C, S())
    close(dx.float().cpu().numpy(), xr.grad.numpy(), 1e-7, "maxpool bwd")


def test_spatial_window_pad_and_crop(L):
    x = RNG.standard_normal((2, 6, 5, 3))
    xd = dev(x)
    for (oh, ow, oy, ox) in [(9, 8, -1, -1), (4, 3, 1, 1), (6, 5, 0, 0), (7, 4, -1, 1)]:
        out = torch.full((2, oh, ow, 3), 9.0).cuda()
        L.spatial_window(xd.data_ptr(), out.data_ptr(), F32, 2, 6, 5, oh, ow, 3, oy, ox, S())
        want = T.spatial_window(torch.as_tensor(x, dtype=torch.float32), oh, ow, oy, ox)
        assert torch.equal(out.cpu(), want), (oh, ow, oy, ox)


def test_dropout_follows_the_philox_contract(L):
    B, shape = 3, (3, 5, 7, 6)
    x = RNG.standard_normal(shape)
    xd = dev(x)
    step = torch.tensor([4], dtype=torch.int32).cuda()
    y = torch.empty_like(xd)
    L.dropout(xd.data_ptr(), y.data_ptr(), F32, int(np.prod(shape[1:])), B, 0.7, 1234567890123, step.data_ptr(), 99, 5, S())
    keep = T.dropout_keep_mask(shape, 0.7, 1234567890123, 4, 99, sample_offset=5)
    want = np.where(keep, x.astype(np.float32) / np.float32(0.7), 0.0)
    close(y.cpu().numpy(), want, 1e-6, "dropout")
    assert 0.6 < keep.mean() < 0.8
    L.dropout(xd.data_ptr(), y.data_ptr(), F32, int(np.prod(shape[1:])), B, 1.0, 1, step.data_ptr(), 0, 0, S())
    assert torch.equal(y, xd)
