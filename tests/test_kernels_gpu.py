"""Per-kernel parity (MI355X): every C-ABI entry point of libphx.so against the CPU oracle
(oracle.tf1_ops restates the TF 1.12 op; torch autograd of the oracle gives the expected gradients).

Tolerances: fp32 kernels 1e-5 relative-to-max (fp32 accumulation order); bf16 storage paths are compared
with an oracle evaluated on the SAME bf16-rounded inputs, so the only differences are fp32 accumulation
order and the final bf16 rounding of the output (2^-8 relative)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import philox
from oracle import tf1_ops as T

pytestmark = pytest.mark.gpu

F32, BF16 = 0, 1


@pytest.fixture(scope="module")
def L():
    from phiseg_code_amd import runtime as rt
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return rt.lib()


@pytest.fixture(scope="module")
def Ld():
    """The test build of the library (libphx_dbg.so, include/phx_debug.h): libphx.so's sources with a SETTABLE kernel-selection policy.
    Only the tests that force a kernel family onto small shapes use it; everything else runs on the product library `L`."""
    from phiseg_code_amd import runtime as rt
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return rt.debug_lib()


@pytest.fixture
def policy(Ld):
    """Kernel-selection policy of the forward / data-gradient launches in the test build (phx_debug_conv_policy /
    phx_debug_pair_kernel_grid), restored to the product library's constants afterwards: policy(large_maps, big_tiles, pair_grid)."""
    def set_(large_maps=1, big_tiles=1, pair_grid=0):
        Ld.debug_conv_policy(large_maps, big_tiles)
        Ld.debug_pair_kernel_grid(pair_grid)
    yield set_
    Ld.debug_conv_policy(1, 1)
    Ld.debug_pair_kernel_grid(0)


def S():
    return torch.cuda.current_stream().cuda_stream


def dev(a, dt=F32):
    t = torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda()
    return t.to(torch.bfloat16).contiguous() if dt == BF16 else t.contiguous()


def host(t):
    torch.cuda.synchronize()
    return t.float().cpu().double().numpy()


def rounded(a, dt):
    """value of `a` after storage in dt (so the oracle sees what the kernel sees)."""
    t = torch.as_tensor(a, dtype=torch.float32)
    return (t.to(torch.bfloat16).float() if dt == BF16 else t).double()


def tdt(dt):
    return torch.bfloat16 if dt == BF16 else torch.float32


def close(got, ref, rel, what=""):
    ref = np.asarray(ref, dtype=np.float64)
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(np.asarray(got, dtype=np.float64) - ref).max() / scale
    assert err <= rel, "%s: rel-to-max err %.3e > %.1e" % (what, err, rel)


RNG = np.random.default_rng(0)


# ------------------------------------------------------------------------------------------------
def test_philox_normal_matches_oracle(L):
    step = torch.tensor([7], dtype=torch.int32).cuda()
    for (B, per, stream, off) in [(3, 2048, 5, 0), (2, 37, 33, 10)]:
        out = torch.empty(B, per, dtype=torch.float32).cuda()
        L.philox_normal(out.data_ptr(), B, per, 42 + (1 << 40), step.data_ptr(), stream, off, S())
        ref = philox.normal(42 + (1 << 40), 7, stream, B, per, sample_offset=off, dtype=np.float64)
        np.testing.assert_allclose(host(out), ref, rtol=0, atol=3e-6)


CONV_DIRECT_CASES = [
    # B, H, W, Cin, Cout, k, act, bias, xdt, ydt
    (2, 16, 16, 3, 32, 3, "relu", False, F32, F32),
    (2, 64, 64, 1, 4, 3, "identity", True, F32, F32),
    (3, 8, 8, 24, 2, 1, "softplus", True, F32, F32),
    (2, 4, 4, 24, 2, 3, "identity", True, F32, F32),
    (5, 2, 2, 24, 24, 3, "relu", False, F32, F32),
    (2, 32, 32, 2, 64, 3, "identity", False, F32, BF16),
    (2, 16, 16, 128, 2, 1, "identity", True, BF16, F32),
    (1, 6, 6, 38, 32, 1, "relu", False, BF16, BF16),
    (2, 3, 3, 20, 17, 3, "identity", True, F32, F32),
    (1, 128, 128, 32, 32, 3, "identity", False, F32, F32),
    # small maps with a narrow side: the wave-per-pixel kernels (top-level 3x3 mu convolution, 192 -> 2 at 2x2, and its backward)
    (64, 2, 2, 192, 2, 3, "identity", True, BF16, F32),
    (7, 3, 5, 70, 3, 3, "softplus", True, BF16, F32),
    (9, 4, 4, 2, 130, 3, "relu", True, F32, BF16),
]
ACT = {"identity": 0, "relu": 1, "softplus": 2}


def oracle_conv(x, w, b, act):
    y = T.conv2d_same(x, w)
    if b is not None:
        y = T.bias_add(y, b)
    return {"identity": lambda v: v, "relu": T.relu, "softplus": T.softplus}[act](y)


@pytest.mark.parametrize("case", CONV_DIRECT_CASES)
def test_conv2d_direct_fwd_dgrad_wgrad(L, case):
    B, H, W, Cin, Cout, k, act, use_bias, xdt, ydt = case
    x = RNG.standard_normal((B, H, W, Cin))
    w = RNG.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)
    b = RNG.standard_normal(Cout) * 0.3
    xr = rounded(x, xdt).requires_grad_(True)
    wr = torch.as_tensor(w, dtype=torch.float32).double().requires_grad_(True)
    br = torch.as_tensor(b, dtype=torch.float32).double().requires_grad_(True)
    yr = oracle_conv(xr, wr, br if use_bias else None, act)
    xd, wd, bd = dev(x, xdt), dev(w), dev(b)
    y = torch.empty(B, H, W, Cout, dtype=tdt(ydt)).cuda()
    stats = torch.zeros(2 * Cout, dtype=torch.float32).cuda()
    L.conv2d_direct(xd.data_ptr(), xdt, wd.data_ptr(), bd.data_ptr() if use_bias else None, y.data_ptr(), ydt,
                    B, H, W, Cin, Cout, k, ACT[act], 0, stats.data_ptr(), S())
    tol = 1e-5 if ydt == F32 else 6e-3
    close(host(y), yr.detach().numpy(), tol, "fwd")
    yst = host(y)
    close(host(stats)[0::2], yst.sum(axis=(0, 1, 2)), 1e-4 if ydt == F32 else 1e-3, "stats sum")
    close(host(stats)[1::2], (yst ** 2).sum(axis=(0, 1, 2)), 1e-4 if ydt == F32 else 1e-3, "stats sumsq")
    y2 = torch.empty_like(y)                    # without statistics (how the heads are launched; small maps take their own kernels)
    L.conv2d_direct(xd.data_ptr(), xdt, wd.data_ptr(), bd.data_ptr() if use_bias else None, y2.data_ptr(), ydt,
                    B, H, W, Cin, Cout, k, ACT[act], 0, None, S())
    close(host(y2), yr.detach().numpy(), tol, "fwd (no stats)")
    # gradients of sum(conv(x, w) * dy) (pre-activation): dgrad via transpose_flip, wgrad kernel
    dy = RNG.standard_normal((B, H, W, Cout))
    dyr = rounded(dy, ydt)
    pre = T.conv2d_same(xr, wr) + (br.reshape(1, 1, 1, -1) if use_bias else 0.0)
    (pre * dyr).sum().backward()
    dyd = dev(dy, ydt)
    dx = torch.empty(B, H, W, Cin, dtype=tdt(xdt)).cuda()
    L.conv2d_direct(dyd.data_ptr(), ydt, wd.data_ptr(), None, dx.data_ptr(), xdt, B, H, W, Cin, Cout, k, 0, 1,
                    None, S())
    close(host(dx), xr.grad.numpy(), 1e-5 if xdt == F32 else 6e-3, "dgrad")
    dw = torch.zeros(k, k, Cin, Cout, dtype=torch.float32).cuda()
    db = torch.zeros(Cout, dtype=torch.float32).cuda()
    L.conv2d_direct_wgrad(xd.data_ptr(), xdt, dyd.data_ptr(), ydt, dw.data_ptr(), db.data_ptr(), B, H, W, Cin,
                          Cout, k, S())
    close(host(dw), wr.grad.numpy(), 2e-5, "wgrad")
    if use_bias:
        close(host(db), br.grad.numpy(), 2e-5, "dbias")
    # the ordered form (pixel slices -> partial filters -> fold in slice order; what PHX_DETERMINISTIC=1 plans launch): the same
    # gradients, accumulated onto what dw / db hold, and bit-identical from call to call
    wsb = int(L.conv2d_direct_wgrad_ordered_ws_bytes(B, H, W, Cin, Cout, k))
    ws = torch.empty(max(wsb // 4, 1), dtype=torch.float32).cuda()
    outs = []
    for _ in range(2):
        dw2 = torch.full((k, k, Cin, Cout), 0.5, dtype=torch.float32).cuda()
        db2 = torch.full((Cout,), 0.25, dtype=torch.float32).cuda()
        L.conv2d_direct_wgrad_ordered(xd.data_ptr(), xdt, dyd.data_ptr(), ydt, dw2.data_ptr(), db2.data_ptr(), ws.data_ptr(), wsb,
                                      B, H, W, Cin, Cout, k, S())
        outs.append((host(dw2), host(db2)))
    close(outs[0][0] - 0.5, wr.grad.numpy(), 2e-5, "wgrad (ordered)")
    if use_bias:
        close(outs[0][1] - 0.25, br.grad.numpy(), 2e-5, "dbias (ordered)")
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


F32_MFMA_CASES = [
    # B, H, W, Cin, Cout, act, bias -- csrc/conv_f32_mfma.hip: every tile geometry (16 x 16, whole small maps with several images per tile,
    # partial tiles of the 192-type maps), narrow inputs (Cin = 1 / 2 / 3 / 6: scalar loader), 32- and 64-wide channel blocks, all four
    # wave layouts of the filter gradient
    (2, 16, 16, 32, 32, "identity", False),
    (1, 32, 32, 64, 64, "relu", True),
    (3, 8, 8, 64, 128, "identity", False),
    (5, 4, 4, 192, 64, "relu", True),
    (40, 2, 2, 32, 96, "identity", False),
    (2, 12, 12, 3, 64, "softplus", True),
    (3, 3, 3, 64, 32, "identity", False),
    (1, 24, 24, 1, 32, "identity", True),
    (2, 6, 6, 2, 192, "relu", False),
    (1, 48, 48, 96, 32, "identity", False),
    (2, 64, 64, 128, 192, "relu", False),
    (1, 128, 128, 32, 32, "identity", True),
    (3, 16, 32, 6, 160, "identity", False),
    # at least 512 blocks of 256 rows: the large-tile kernel (everything above takes the 64-row tiles whose waves split the reduction)
    (8, 128, 128, 32, 64, "relu", True),
    (32, 64, 64, 3, 32, "identity", False),
    (57, 48, 48, 8, 64, "identity", True),
    (130, 16, 16, 64, 96, "relu", False),
    # ... and its LDS-DMA form with loader waves: 32-channel blocks (N % 64 != 0), and a block count that leaves 512 slots half empty
    (8, 128, 128, 64, 32, "relu", True),
    (9, 128, 64, 12, 96, "identity", False),
    (24, 64, 64, 32, 128, "identity", True),
]


@pytest.mark.parametrize("case", F32_MFMA_CASES)
def test_conv3x3_f32_mfma_fwd_dgrad_wgrad(L, case):
    """The fp32 matrix-instruction convolution of the fp32 parity path (v_mfma_f32_32x32x2_f32: an fp32 FMA chain) against the oracle's
    conv2d_same and its autograd in float64 -- forward with bias / activation, data gradient (the same kernel on the flipped,
    transposed packed filter), filter and bias gradient (accumulated onto what dw / db hold; bit-identical from call to call: the
    partial filters are summed in slice order)."""
    B, H, W, Cin, Cout, act, use_bias = case
    x = RNG.standard_normal((B, H, W, Cin))
    w = RNG.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)
    b = RNG.standard_normal(Cout) * 0.3
    xr = rounded(x, F32).requires_grad_(True)
    wr = torch.as_tensor(w, dtype=torch.float32).double().requires_grad_(True)
    br = torch.as_tensor(b, dtype=torch.float32).double().requires_grad_(True)
    yr = oracle_conv(xr, wr, br if use_bias else None, act)
    xd, wd, bd = dev(x), dev(w), dev(b)
    assert L.conv3x3_f32_mfma_supported(B, H, W, Cin, Cout)
    wf = torch.empty(int(L.conv3x3_f32_mfma_packed_floats(Cin, Cout)), dtype=torch.float32).cuda()
    dgrad = Cin % 32 == 0
    wdg = torch.empty(int(L.conv3x3_f32_mfma_packed_floats(Cout, Cin)), dtype=torch.float32).cuda() if dgrad else None
    rec = np.zeros(1, dtype=[("w", "<u8"), ("wf", "<u8"), ("wd", "<u8"), ("cin", "<i4"), ("cout", "<i4")])
    rec[0] = (wd.data_ptr(), wf.data_ptr(), wdg.data_ptr() if dgrad else 0, Cin, Cout)
    desc = torch.from_numpy(rec.view(np.uint8).copy()).cuda()
    L.pack_conv3x3_f32_multi(desc.data_ptr(), 1, S())
    y = torch.empty(B, H, W, Cout, dtype=torch.float32).cuda()
    L.conv3x3_f32_mfma(xd.data_ptr(), wf.data_ptr(), bd.data_ptr() if use_bias else None, y.data_ptr(), B, H, W, Cin, Cout, ACT[act], S())
    close(host(y), yr.detach().numpy(), 1e-5, "fwd")
    dy = RNG.standard_normal((B, H, W, Cout))
    dyr = rounded(dy, F32)
    pre = T.conv2d_same(xr, wr) + (br.reshape(1, 1, 1, -1) if use_bias else 0.0)
    (pre * dyr).sum().backward()
    dyd = dev(dy)
    if dgrad:
        dx = torch.empty(B, H, W, Cin, dtype=torch.float32).cuda()
        L.conv3x3_f32_mfma(dyd.data_ptr(), wdg.data_ptr(), None, dx.data_ptr(), B, H, W, Cout, Cin, 0, S())
        close(host(dx), xr.grad.numpy(), 1e-5, "dgrad")
    assert L.conv3x3_f32_mfma_wgrad_supported(B, H, W, Cin, Cout)
    wsb = int(L.conv3x3_f32_mfma_wgrad_ws_bytes(B, H, W, Cin, Cout, 1 if use_bias else 0))
    ws = torch.empty(max(wsb // 4, 1), dtype=torch.float32).cuda()
    outs = []
    for _ in range(2):
        dw = torch.full((3, 3, Cin, Cout), 0.5, dtype=torch.float32).cuda()
        db = torch.full((Cout,), 0.25, dtype=torch.float32).cuda()
        L.conv3x3_f32_mfma_wgrad(xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), db.data_ptr() if use_bias else None, ws.data_ptr(), wsb,
                                 B, H, W, Cin, Cout, S())
        outs.append((host(dw), host(db)))
    close(outs[0][0] - 0.5, wr.grad.numpy(), 2e-5, "wgrad")
    if use_bias:
        close(outs[0][1] - 0.25, br.grad.numpy(), 2e-5, "dbias")
    else:
        assert np.all(outs[0][1] == 0.25)
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


MFMA_CASES = [
    # B, H, W, K(Cin), N(Cout)
    (2, 16, 16, 32, 32),
    (1, 32, 32, 64, 64),
    (3, 8, 8, 64, 128),
    (5, 4, 4, 192, 64),
    (40, 2, 2, 32, 96),
    (2, 12, 12, 32, 64),
    (3, 3, 3, 64, 32),
    (1, 128, 128, 32, 32),
    (2, 64, 64, 128, 192),
    (1, 48, 48, 96, 32),
]


@pytest.mark.parametrize("case", MFMA_CASES)
def test_conv3x3_mfma_fwd_dgrad_wgrad(L, case):
    _mfma_case(L, case)


# the 16 x 32-pixel-tile / 8-wave forward kernels (chosen by policy only for large maps): forced here on small ones
@pytest.mark.parametrize("case", [(2, 32, 32, 64, 128), (1, 64, 32, 32, 32), (3, 32, 16, 96, 64), (1, 32, 48, 32, 256)])
def test_conv3x3_mfma_big_tiles(Ld, case, policy):
    L = Ld                                   # (the test build: this test sets the kernel policy)
    policy(big_tiles=2)
    _mfma_case(L, case)


@pytest.mark.parametrize("case", [(2, 16, 16, 38, 32), (1, 32, 32, 64, 64)])
def test_conv1x1_as_centre_tap(L, case):
    """1x1 filters (prob_unet2D's recombination layers, model_zoo/likelihoods.py) run on the 3x3 MFMA kernels as the centre
    tap, input channels zero-padded to a multiple of 32: forward, data gradient and filter gradient vs the 1x1 oracle."""
    B, H, W, C, N = case
    Cp = (C + 31) // 32 * 32
    x = RNG.standard_normal((B, H, W, C))
    w = RNG.standard_normal((1, 1, C, N)) / np.sqrt(C)
    xr = rounded(x, BF16).requires_grad_(True)
    wr = rounded(w, BF16).requires_grad_(True)
    xd, wd = dev(x, BF16), dev(w)
    xp = torch.empty(B, H, W, Cp, dtype=torch.bfloat16).cuda()
    L.pad_channels_bf16(xd.data_ptr(), BF16, C, xp.data_ptr(), Cp, B * H * W, S())
    wf = torch.empty(9 * N * Cp, dtype=torch.bfloat16).cuda()
    wg = torch.empty(9 * N * Cp, dtype=torch.bfloat16).cuda()
    desc = np.zeros(1, dtype=[("w", "<u8"), ("wf", "<u8"), ("wd", "<u8"), ("cin", "<i4"), ("cpad", "<i4"), ("cout", "<i4"),
                              ("k1", "<i4")])
    desc[0] = (wd.data_ptr(), wf.data_ptr(), wg.data_ptr(), C, Cp, N, 1)
    dd = torch.from_numpy(desc.view(np.uint8).copy()).cuda()
    L.pack_conv3x3_bf16_multi(dd.data_ptr(), 1, S())
    y = torch.empty(B, H, W, N, dtype=torch.bfloat16).cuda()
    L.conv3x3_mfma_bf16(xp.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, None, B, H, W, Cp, N, S())
    yr = torch.einsum("bhwc,cn->bhwn", xr, wr[0, 0])
    close(host(y), yr.detach().numpy(), 6e-3, "1x1 fwd")
    dy = RNG.standard_normal((B, H, W, N))
    dyr = rounded(dy, BF16)
    (yr * dyr).sum().backward()
    dyd = dev(dy, BF16)
    dxp = torch.empty(B, H, W, Cp, dtype=torch.bfloat16).cuda()
    L.conv3x3_mfma_bf16(dyd.data_ptr(), wg.data_ptr(), dxp.data_ptr(), None, 0, None, B, H, W, N, Cp, S())
    close(host(dxp)[..., :C], xr.grad.numpy(), 6e-3, "1x1 dgrad")
    if Cp > C:
        assert np.abs(host(dxp)[..., C:]).max() == 0.0
    dwp = torch.zeros(9 * Cp * N, dtype=torch.float32).cuda()
    L.conv3x3_wgrad_mfma_bf16(xp.data_ptr(), dyd.data_ptr(), dwp.data_ptr(), None, 0, B, H, W, Cp, N, S())
    dw = torch.zeros(C, N, dtype=torch.float32).cuda()
    L.unpad_filter_grad_center(dwp.data_ptr(), dw.data_ptr(), C, Cp, N, S())
    close(host(dw), wr.grad.numpy()[0, 0], 1e-4, "1x1 wgrad")


@pytest.mark.parametrize("case", [(64, 8, 8, 192, 192), (64, 16, 16, 64, 96), (3, 8, 8, 32, 32), (9, 4, 4, 64, 64), (64, 16, 16, 384, 192)])
def test_conv3x3_mfma_statistics_by_atomics(L, case):
    """phx_conv3x3_mfma_bf16_stats_atomic: the convolution of a layer with few pixel tiles adds {sum y, sum y^2} of its (bf16-rounded)
    output straight into sums[N][2], the accumulator phx_norm_apply_fused reads -- the batch-norm layers of the H <= 16 levels then
    need neither a statistics pass over y nor a reduction launch."""
    B, H, W, K, N = case
    assert L.conv3x3_mfma_stats_atomic_supported(B, H, W, K, N) == 1
    assert L.conv3x3_mfma_stats_atomic_supported(64, 128, 128, 32, 32) == 0          # thousands of tiles: partial sums + reduction
    x = RNG.standard_normal((B, H, W, K))
    w = RNG.standard_normal((3, 3, K, N)) / np.sqrt(9 * K)
    xd, wd = dev(x, BF16), dev(w)
    wf = torch.empty(9 * N * K, dtype=torch.bfloat16).cuda()
    wg = torch.empty(9 * N * K, dtype=torch.bfloat16).cuda()
    L.pack_conv3x3_bf16(wd.data_ptr(), wf.data_ptr(), wg.data_ptr(), K, N, S())
    y0 = torch.empty(B, H, W, N, dtype=torch.bfloat16).cuda()
    L.conv3x3_mfma_bf16_ws(xd.data_ptr(), wf.data_ptr(), y0.data_ptr(), None, 0, None, None, 0, B, H, W, K, N, S())
    y = torch.empty_like(y0)
    sums = torch.full((N, 2), 0.5, dtype=torch.float32).cuda()                        # accumulated (+=)
    L.conv3x3_mfma_bf16_stats_atomic(xd.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, sums.data_ptr(), B, H, W, K, N, S())
    assert torch.equal(y, y0)
    yf = host(y).reshape(-1, N).astype(np.float64)
    got = host(sums) - 0.5
    close(got[:, 0], yf.sum(0), 2e-5, "sum y")
    close(got[:, 1], (yf ** 2).sum(0), 2e-5, "sum y^2")
    from phiseg_code_amd.runtime import PhxError
    with pytest.raises(PhxError):
        L.conv3x3_mfma_bf16_stats_atomic(xd.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, None, B, H, W, K, N, S())


@pytest.mark.parametrize("case", [(64, 2, 2, 192, 192, 0), (64, 4, 4, 192, 192, 0), (12, 4, 4, 256, 192, 64), (12, 2, 2, 32, 64, 0), (12, 8, 8, 192, 192, 0),
                                  (5, 3, 3, 64, 96, 0), (64, 4, 4, 32, 192, 0), (3, 16, 16, 384, 64, 192)])
def test_conv3x3_mfma_fp32_output(L, case):
    """phx_conv3x3_mfma_bf16_f32out (round 5): the plain convolution of a small map with its fp32 accumulators written out unrounded --
    split-K slices summed by the finishing pass, a single slice written directly, plain and concat-free input -- against the oracle
    on the bf16-rounded operands: 2e-5 of the output's range where the bf16 tensor is good to 4e-3."""
    B, H, W, K, N, K1 = case
    assert L.conv3x3_mfma_f32out_supported(B, H, W, K, N) == 1 and L.conv3x3_mfma_f32out_supported(64, 128, 128, 128, 128) == 0
    x = RNG.standard_normal((B, H, W, K))
    w = RNG.standard_normal((3, 3, K, N)) / np.sqrt(9 * K)
    xd, wd = dev(x, BF16), dev(w)
    wf = torch.empty(9 * N * K, dtype=torch.bfloat16).cuda()
    L.pack_conv3x3_bf16(wd.data_ptr(), wf.data_ptr(), None, K, N, S())
    ref = T.conv2d_same(rounded(x, BF16), rounded(w, BF16)).numpy()
    wsb = int(L.conv3x3_mfma_ws_bytes(B, H, W, K, N))
    assert (wsb > 0) == (int(L.conv3x3_mfma_ksplit(B, H, W, K, N)) > 1)
    ws = torch.empty(max(wsb // 4, 1), dtype=torch.float32).cuda()
    y = torch.full((B, H, W, N), 7.0, dtype=torch.float32).cuda()
    if K1:
        xa, xb = dev(x[..., :K1], BF16), dev(x[..., K1:], BF16)
        L.conv3x3_mfma_bf16_f32out(xa.data_ptr(), xb.data_ptr(), K1, wf.data_ptr(), y.data_ptr(), 1, ws.data_ptr() if wsb else None, wsb,
                                   B, H, W, K, N, S())
    else:
        L.conv3x3_mfma_bf16_f32out(xd.data_ptr(), None, 0, wf.data_ptr(), y.data_ptr(), 1, ws.data_ptr() if wsb else None, wsb, B, H, W, K, N, S())
    close(host(y), ref, 2e-5, "fp32-output convolution")
    nz = int(L.conv3x3_mfma_ksplit(B, H, W, K, N))
    if nz > 1 and not K1:            # sum_slices = 0: the slices stay in the workspace, in the order the finishing pass adds them
        ws.zero_()
        y2 = torch.full_like(y, 7.0)
        L.conv3x3_mfma_bf16_f32out(xd.data_ptr(), None, 0, wf.data_ptr(), y2.data_ptr(), 0, ws.data_ptr(), wsb, B, H, W, K, N, S())
        sl = ws[:nz * B * H * W * N].view(nz, B, H, W, N)
        acc = sl[0].clone()
        for z in range(1, nz):
            acc += sl[z]
        assert torch.equal(acc, y) and float(y2.min()) == 7.0          # bit-identical sum; y_f32 untouched
    from phiseg_code_amd.runtime import PhxError
    if wsb:
        with pytest.raises(PhxError):            # a split-K shape without its workspace
            L.conv3x3_mfma_bf16_f32out(xd.data_ptr(), None, 0, wf.data_ptr(), y.data_ptr(), 1, None, 0, B, H, W, K, N, S())


def _mfma_case(L, case):
    B, H, W, K, N = case
    x = RNG.standard_normal((B, H, W, K))
    w = RNG.standard_normal((3, 3, K, N)) / np.sqrt(9 * K)
    b = RNG.standard_normal(N) * 0.3
    xr = rounded(x, BF16).requires_grad_(True)
    wr = rounded(w, BF16).requires_grad_(True)       # the packed filter is bf16
    xd, wd, bd = dev(x, BF16), dev(w), dev(b)
    wf = torch.empty(9 * N * K, dtype=torch.bfloat16).cuda()
    wg = torch.empty(9 * N * K, dtype=torch.bfloat16).cuda()
    L.pack_conv3x3_bf16(wd.data_ptr(), wf.data_ptr(), wg.data_ptr(), K, N, S())
    ntile = L.conv3x3_mfma_bf16_tiles(B, H, W, K, N)
    part = torch.zeros(ntile, 2, N, dtype=torch.float32).cuda()
    y = torch.empty(B, H, W, N, dtype=torch.bfloat16).cuda()
    L.conv3x3_mfma_bf16(xd.data_ptr(), wf.data_ptr(), y.data_ptr(), bd.data_ptr(), 1, part.data_ptr(), B, H, W, K,
                        N, S())
    yr = oracle_conv(xr, wr, torch.as_tensor(b, dtype=torch.float32).double(), "relu")
    close(host(y), yr.detach().numpy(), 6e-3, "mfma fwd")
    yst = host(y)
    sums = torch.zeros(N, 2, dtype=torch.float32).cuda()
    L.norm_reduce_partials(part.data_ptr(), ntile, N, sums.data_ptr(), S())
    close(host(sums)[:, 0], yst.sum(axis=(0, 1, 2)), 1e-3, "epilogue sum")
    close(host(sums)[:, 1], (yst ** 2).sum(axis=(0, 1, 2)), 1e-3, "epilogue sumsq")
    # no-bias identity
    L.conv3x3_mfma_bf16(xd.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, None, B, H, W, K, N, S())
    pre = T.conv2d_same(xr, wr)
    close(host(y), pre.detach().numpy(), 6e-3, "mfma fwd id")
    # split-K variant (used when the shape has few pixel tiles; same result, with and without bias / activation)
    wsb = int(L.conv3x3_mfma_ws_bytes(B, H, W, K, N))
    wsk = torch.empty(max(wsb // 4, 1), dtype=torch.float32).cuda()
    y2 = torch.empty(B, H, W, N, dtype=torch.bfloat16).cuda()
    L.conv3x3_mfma_bf16_ws(xd.data_ptr(), wf.data_ptr(), y2.data_ptr(), None, 0, None, wsk.data_ptr(), wsb, B, H, W, K, N, S())
    close(host(y2), pre.detach().numpy(), 6e-3, "mfma fwd id (split-K path)")
    L.conv3x3_mfma_bf16_ws(xd.data_ptr(), wf.data_ptr(), y2.data_ptr(), bd.data_ptr(), 1, None, wsk.data_ptr(), wsb, B, H, W, K, N,
                           S())
    close(host(y2), yr.detach().numpy(), 6e-3, "mfma fwd bias relu (split-K path)")
    dy = RNG.standard_normal((B, H, W, N))
    dyr = rounded(dy, BF16)
    (pre * dyr).sum().backward()
    dyd = dev(dy, BF16)
    dx = torch.empty(B, H, W, K, dtype=torch.bfloat16).cuda()
    L.conv3x3_mfma_bf16(dyd.data_ptr(), wg.data_ptr(), dx.data_ptr(), None, 0, None, B, H, W, N, K, S())
    close(host(dx), xr.grad.numpy(), 6e-3, "mfma dgrad")
    dw = torch.zeros(3, 3, K, N, dtype=torch.float32).cuda()
    L.conv3x3_wgrad_mfma_bf16(xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), None, 0, B, H, W, K, N, S())
    close(host(dw), wr.grad.numpy(), 1e-4, "mfma wgrad (atomics)")
    wsb = int(L.conv3x3_wgrad_ws_bytes(B, H, W, K, N))
    ws = torch.empty(wsb // 4, dtype=torch.float32).cuda()
    dw2 = torch.zeros(3, 3, K, N, dtype=torch.float32).cuda()
    L.conv3x3_wgrad_mfma_bf16(xd.data_ptr(), dyd.data_ptr(), dw2.data_ptr(), ws.data_ptr(), wsb, B, H, W, K, N, S())
    close(host(dw2), wr.grad.numpy(), 1e-4, "mfma wgrad (workspace)")


@pytest.mark.parametrize("case", [(3, 16, 32), (2, 48, 64), (5, 32, 32), (1, 16, 96), (17, 128, 128)])
def test_conv3x3_32_channel_layers_on_the_pre_normalisation_tensor(Ld, case, policy):
    """phx_conv3x3_mfma_bf16_xf / phx_conv3x3_wgrad_mfma_bf16_partial_xf (round 5): conv2d -> batch_norm -> relu -> conv2d
    (tfwrapper/layers.py:123-135) on the 32-channel large-map layers WITHOUT the activation tensor -- k_conv3x3_c32 and the LDS-DMA filter
    gradient re-form a = relu(x * scale + shift) in place in their staged patches (image edges on every side: the zero padding is of a,
    not of x; odd tile counts).  Bit-identical to the same launches on the bf16 activation phx_affine_act materialises, and against the
    oracle's conv2d of relu(x * scale + shift) on the small cases."""
    L = Ld
    B, H, W = case
    K = N = 32
    policy(large_maps=2)
    assert L.conv3x3_xf_supported(B, H, W, K, N) == 1 and L.conv3x3_xf_supported(B, H, W, 64, 64) == 0
    x = RNG.standard_normal((B, H, W, K)).astype(np.float32)
    w = RNG.standard_normal((3, 3, K, N)) / np.sqrt(9 * K)
    sc, sh = 1.0 + 0.3 * RNG.standard_normal(K), 0.2 * RNG.standard_normal(K)
    xd, wd, scd, shd = dev(x, BF16), dev(w), dev(sc), dev(sh)
    P = B * H * W
    a = torch.empty_like(xd)
    L.affine_act(xd.data_ptr(), BF16, scd.data_ptr(), shd.data_ptr(), a.data_ptr(), BF16, 1, P, K, 1, S())
    wf = torch.empty(9 * N * K, dtype=torch.bfloat16).cuda()
    L.pack_conv3x3_bf16(wd.data_ptr(), wf.data_ptr(), None, K, N, S())
    nt = L.conv3x3_mfma_bf16_tiles(B, H, W, K, N)
    y1, y2 = (torch.empty(B, H, W, N, dtype=torch.bfloat16).cuda() for _ in range(2))
    p1, p2 = (torch.zeros(nt, 2, N, dtype=torch.float32).cuda() for _ in range(2))
    L.conv3x3_mfma_bf16(a.data_ptr(), wf.data_ptr(), y1.data_ptr(), None, 0, p1.data_ptr(), B, H, W, K, N, S())
    L.conv3x3_mfma_bf16_xf(xd.data_ptr(), scd.data_ptr(), shd.data_ptr(), wf.data_ptr(), y2.data_ptr(), p2.data_ptr(), B, H, W, K, N, S())
    assert torch.equal(y1, y2) and torch.equal(p1, p2)
    if P <= 65536:
        ar = torch.relu(rounded(x, BF16) * torch.as_tensor(sc, dtype=torch.float32).double() + torch.as_tensor(sh, dtype=torch.float32).double())
        ref = T.conv2d_same(rounded(ar.float().numpy(), BF16), rounded(w, BF16))
        close(host(y2), ref.numpy(), 8e-3, "conv on the transformed input")
    if not L.conv3x3_wgrad_xf_supported(B, H, W, K, N):
        assert B < 17
        return
    dy = dev(RNG.standard_normal((B, H, W, N)).astype(np.float32), BF16)
    wsb = int(L.conv3x3_wgrad_ws_bytes(B, H, W, K, N))
    ws1, ws2 = (torch.zeros(wsb // 4, dtype=torch.float32).cuda() for _ in range(2))
    dw = torch.zeros(3, 3, K, N, dtype=torch.float32).cuda()
    L.conv3x3_wgrad_mfma_bf16_partial(a.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws1.data_ptr(), wsb, B, H, W, K, N, S())
    L.conv3x3_wgrad_mfma_bf16_partial_xf(xd.data_ptr(), scd.data_ptr(), shd.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws2.data_ptr(), wsb, B, H, W, K, N, S())
    torch.cuda.synchronize()
    assert torch.equal(ws1, ws2) and float(ws1.abs().max()) > 0


# the anti-phase filter-gradient kernel (k_conv3x3_wgrad_pp: 64 x 64 channel blocks on 16 x 16 pixel tiles): one tile (half B idle), odd
# and even tile counts per block, maps that are not multiples of the tile, several channel blocks, the two-tensor (concat-free) input
@pytest.mark.parametrize("case", [(5, 16, 16, 64, 64, 0), (3, 16, 32, 64, 64, 0), (6, 16, 16, 64, 128, 0), (2, 40, 24, 128, 64, 0),
                                  (9, 48, 48, 192, 192, 0), (2, 64, 64, 192, 192, 0), (3, 32, 32, 128, 64, 64), (67, 16, 16, 64, 64, 0)])
def test_conv3x3_wgrad_anti_phase_64x64(L, case):
    B, H, W, K, N, K1 = case
    x, dy = RNG.standard_normal((B, H, W, K)), RNG.standard_normal((B, H, W, N))
    xr = rounded(x, BF16)
    wr = torch.zeros(3, 3, K, N, dtype=torch.float64, requires_grad=True)
    (T.conv2d_same(xr, wr) * rounded(dy, BF16)).sum().backward()
    dyd = dev(dy, BF16)
    wsb = int(L.conv3x3_wgrad_ws_bytes_dual(B, H, W, K, N, K1))
    plan = (ctypes.c_int * 6)()
    L.conv3x3_wgrad_reduce_plan_dual(B, H, W, K, N, K1, plan)
    assert plan[0] and plan[2] == 64 and plan[3] == 64          # workspace path, 64 x 64 channel blocks
    ws = torch.empty(wsb // 4, dtype=torch.float32).cuda()
    dw = torch.full((3, 3, K, N), 0.5, dtype=torch.float32).cuda()      # accumulate semantics
    if K1:
        xa, xb = dev(x[..., :K1], BF16), dev(x[..., K1:], BF16)
        L.conv3x3_wgrad_mfma_bf16_dual(xa.data_ptr(), xb.data_ptr(), K1, dyd.data_ptr(), dw.data_ptr(), ws.data_ptr(), wsb, B, H, W, K, N, 1, S())
    else:
        xd = dev(x, BF16)
        L.conv3x3_wgrad_mfma_bf16(xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), ws.data_ptr(), wsb, B, H, W, K, N, S())
    close(host(dw) - 0.5, wr.grad.numpy(), 1e-4, "anti-phase filter gradient")


@pytest.mark.parametrize("case", [(64, 128, 128, 128, 128), (64, 64, 64, 192, 192), (64, 128, 128, 64, 128), (64, 128, 128, 192, 32),
                                  (64, 128, 128, 32, 32), (64, 32, 32, 128, 128), (64, 16, 16, 192, 192)])
def test_conv3x3_forward_full_size_vs_direct_kernel_and_statistics(L, case):
    """BASELINE-size forward launches (the default policy's kernels: pair kernel, k_conv3x3_c32, 256-pixel kernels): the bf16 output
    and the epilogue's per-channel statistics against the fp32 direct kernel (itself checked against the oracle above) on the same bf16
    input and bf16-rounded filter -- every image of the batch, on the device; the data gradient is the same launch with the flipped
    pack.  (test_conv3x3_full_size_against_the_cpu_oracle below compares the same launches with the oracle directly.)"""
    B, H, W, K, N = case
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.relu(torch.randn(B, H, W, K, device="cuda", generator=g)).to(torch.bfloat16)
    w = (torch.randn(3, 3, K, N, device="cuda", generator=g) / np.sqrt(9 * K)).to(torch.bfloat16).float().contiguous()
    wf = torch.empty(9 * N * K, dtype=torch.bfloat16, device="cuda")
    wg = torch.empty(9 * N * K, dtype=torch.bfloat16, device="cuda")
    L.pack_conv3x3_bf16(w.data_ptr(), wf.data_ptr(), wg.data_ptr(), K, N, S())
    ntile = L.conv3x3_mfma_bf16_tiles(B, H, W, K, N)
    part = torch.zeros(ntile, 2, N, dtype=torch.float32, device="cuda")
    y = torch.empty(B, H, W, N, dtype=torch.bfloat16, device="cuda")
    L.conv3x3_mfma_bf16(x.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, part.data_ptr(), B, H, W, K, N, S())
    ref = torch.empty(B, H, W, N, dtype=torch.float32, device="cuda")
    L.conv2d_direct(x.data_ptr(), BF16, w.data_ptr(), None, ref.data_ptr(), F32, B, H, W, K, N, 3, 0, 0, None, S())
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    assert float((y.float() - ref).abs().max()) <= 6e-3 * scale            # bf16 output rounding (2^-9 of the value, values up to `scale`)
    sums = torch.zeros(N, 2, dtype=torch.float32, device="cuda")
    L.norm_reduce_partials(part.data_ptr(), ntile, N, sums.data_ptr(), S())
    torch.cuda.synchronize()
    yf = y.float().reshape(-1, N).double()
    close(host(sums)[:, 0], yf.sum(0).cpu().numpy(), 1e-4, "epilogue sum (of the stored bf16 values)")
    close(host(sums)[:, 1], (yf * yf).sum(0).cpu().numpy(), 1e-4, "epilogue sum of squares")
    # data gradient: dx = conv(dy, flipped / transposed filter) -- the direct kernel's dgrad mode on the same tensors
    dy = (torch.randn(B, H, W, N, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    dx = torch.empty(B, H, W, K, dtype=torch.bfloat16, device="cuda")
    L.conv3x3_mfma_bf16(dy.data_ptr(), wg.data_ptr(), dx.data_ptr(), None, 0, None, B, H, W, N, K, S())
    dref = torch.empty(B, H, W, K, dtype=torch.float32, device="cuda")
    L.conv2d_direct(dy.data_ptr(), BF16, w.data_ptr(), None, dref.data_ptr(), F32, B, H, W, K, N, 3, 0, 1, None, S())
    torch.cuda.synchronize()
    assert float((dx.float() - dref).abs().max()) <= 6e-3 * float(dref.abs().max())


@pytest.mark.parametrize("case", [(64, 128, 128, 128, 128), (64, 64, 64, 192, 192), (64, 128, 128, 192, 32), (64, 128, 128, 32, 32)])
def test_conv3x3_full_size_against_the_cpu_oracle(L, case):
    """The same BASELINE-size launches against the ORACLE ITSELF (oracle/tf1_ops.py conv2d_same = tf.nn.conv2d 3x3 SAME,
    tfwrapper/layers.py:123, in fp32 on the host; a few seconds per shape with the host's threads, so no device kernel stands between
    the MFMA kernels and the oracle at the sizes the benchmark runs): forward and data gradient on the first and the last eight images
    of the batch-64 launch (first / last pixel tiles and XCD bands), the filter gradient -- a sum over all 64 images -- on the whole
    batch through the oracle's autograd.  Same bf16 inputs and bf16-rounded filter on both sides; tolerances: the bf16 rounding of the
    stored outputs (2^-9 of a value <= `scale`), 1e-4 of the largest entry for the fp32 filter gradient."""
    from oracle import tf1_ops as T
    B, H, W, K, N = case
    g = torch.Generator(device="cuda").manual_seed(23)
    x = torch.relu(torch.randn(B, H, W, K, device="cuda", generator=g)).to(torch.bfloat16)
    dy = (torch.randn(B, H, W, N, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    w = (torch.randn(3, 3, K, N, device="cuda", generator=g) / np.sqrt(9 * K)).to(torch.bfloat16).float().contiguous()
    wf = torch.empty(9 * N * K, dtype=torch.bfloat16, device="cuda")
    wg = torch.empty(9 * N * K, dtype=torch.bfloat16, device="cuda")
    L.pack_conv3x3_bf16(w.data_ptr(), wf.data_ptr(), wg.data_ptr(), K, N, S())
    y = torch.empty(B, H, W, N, dtype=torch.bfloat16, device="cuda")
    dx = torch.empty(B, H, W, K, dtype=torch.bfloat16, device="cuda")
    L.conv3x3_mfma_bf16(x.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, None, B, H, W, K, N, S())
    L.conv3x3_mfma_bf16(dy.data_ptr(), wg.data_ptr(), dx.data_ptr(), None, 0, None, B, H, W, N, K, S())
    wsb = int(L.conv3x3_wgrad_ws_bytes(B, H, W, K, N))
    ws = torch.empty(max(wsb // 4, 1), dtype=torch.float32, device="cuda")
    dw = torch.zeros(3, 3, K, N, dtype=torch.float32, device="cuda")
    L.conv3x3_wgrad_mfma_bf16(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr() if wsb else None, wsb, B, H, W, K, N, S())
    torch.cuda.synchronize()
    xc, dyc, wc = x.float().cpu(), dy.float().cpu(), w.cpu()
    for sl in (slice(0, 8), slice(B - 8, B)):
        xs = xc[sl].clone().requires_grad_(True)
        ref = T.conv2d_same(xs, wc)
        ref.backward(dyc[sl])
        close(y[sl].float().cpu().numpy(), ref.detach().numpy(), 6e-3, "forward, images %s" % (sl,))
        close(dx[sl].float().cpu().numpy(), xs.grad.numpy(), 6e-3, "data gradient, images %s" % (sl,))
    wr = wc.clone().requires_grad_(True)
    for b0 in range(0, B, 16):                                 # (the oracle in four batches of 16: 0.5 GB of host memory at a time)
        T.conv2d_same(xc[b0:b0 + 16], wr).backward(dyc[b0:b0 + 16])
    close(host(dw), wr.grad.numpy(), 1e-4, "filter gradient")


@pytest.mark.parametrize("case", [(64, 128, 128, 128, 128), (64, 64, 64, 192, 192), (64, 128, 128, 32, 32), (64, 16, 16, 192, 192),
                                  (64, 4, 4, 192, 192)])
def test_conv3x3_full_size_adjoint_identities(L, case):
    """The three MFMA launches of a layer at BASELINE sizes are each other's adjoints (no reference kernel involved):
    <conv(x, w), dy> = <x, dgrad(dy, w)> = <w, wgrad(x, dy)> -- forward, data gradient (flipped pack) and filter gradient agree on one
    number up to the bf16 rounding of y and dx, which averages out over >= 10^6 terms."""
    B, H, W, K, N = case
    g = torch.Generator(device="cuda").manual_seed(13)
    x = torch.randn(B, H, W, K, device="cuda", generator=g).to(torch.bfloat16)
    dy = torch.randn(B, H, W, N, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(3, 3, K, N, device="cuda", generator=g) / np.sqrt(9 * K)).to(torch.bfloat16).float().contiguous()
    wf = torch.empty(9 * N * K, dtype=torch.bfloat16, device="cuda")
    wg = torch.empty(9 * N * K, dtype=torch.bfloat16, device="cuda")
    L.pack_conv3x3_bf16(w.data_ptr(), wf.data_ptr(), wg.data_ptr(), K, N, S())

    def conv(src, pack, k, n):
        nb = int(L.conv3x3_mfma_ws_bytes(B, H, W, k, n))
        ws = torch.empty(max(nb // 4, 1), dtype=torch.float32, device="cuda")
        out = torch.empty(B, H, W, n, dtype=torch.bfloat16, device="cuda")
        L.conv3x3_mfma_bf16_ws(src.data_ptr(), pack.data_ptr(), out.data_ptr(), None, 0, None, ws.data_ptr() if nb else None, nb, B, H, W,
                               k, n, S())
        torch.cuda.synchronize()
        return out
    y, dx = conv(x, wf, K, N), conv(dy, wg, N, K)
    wsb = int(L.conv3x3_wgrad_ws_bytes(B, H, W, K, N))
    ws = torch.empty(max(wsb // 4, 1), dtype=torch.float32, device="cuda")
    dw = torch.zeros(3, 3, K, N, dtype=torch.float32, device="cuda")
    L.conv3x3_wgrad_mfma_bf16(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr() if wsb else None, wsb, B, H, W, K, N, S())
    torch.cuda.synchronize()
    a = float((y.double() * dy.double()).sum())
    b = float((dx.double() * x.double()).sum())
    c = float((dw.double() * w.double()).sum())
    # the bf16 rounding of y (relative 2^-9 / sqrt(3) per element, independent) moves <y, dy> by ~1.1e-3 |y| |dy| / sqrt(n); likewise dx.
    # Measured: -3187.2 / -3161.3 / -3161.0 on 128 -> 128 @ 128 x 128 (bound 92), 108.28 / 107.85 / 107.63 on 192 -> 192 @ 4 x 4 (bound 2.9);
    # a launch that dropped one of the nine taps would be off by a ninth of the inner product's own spread |y| |dy| / sqrt(n) ~ 10^4.
    tol = 8e-3 * float(np.sqrt(float((y.double() ** 2).sum()) * float((dy.double() ** 2).sum()))) / np.sqrt(y.numel())
    print("adjoint identities %s: <y, dy> %.6e  <dx, x> %.6e  <dw, w> %.6e  (bound %.3e)" % (case, a, b, c, tol))
    assert abs(a - c) <= tol and abs(b - c) <= tol, (a, b, c, tol)


@pytest.mark.parametrize("case", [(64, 128, 128, 128, 128), (64, 64, 64, 192, 192)])
def test_conv3x3_wgrad_full_size_batch_additivity_and_direct_kernel(L, case):
    """BASELINE-size filter gradients (the shapes bench.py's roofline names), where the oracle takes minutes: (1) against the fp32
    direct kernel (itself checked against the oracle above) on the same bf16 tensors, (2) additivity over the batch -- the gradient of
    the 64 images equals the sum of the gradients of the two halves (other pixel-slice plans, other partial-filter counts)."""
    B, H, W, K, N = case
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.relu(torch.randn(B, H, W, K, device="cuda", generator=g)).to(torch.bfloat16)
    dy = (torch.randn(B, H, W, N, device="cuda", generator=g) * 0.1).to(torch.bfloat16)

    def wgrad(xb, dyb):
        b = xb.shape[0]
        wsb = int(L.conv3x3_wgrad_ws_bytes(b, H, W, K, N))
        ws = torch.empty(max(wsb // 4, 1), dtype=torch.float32, device="cuda")
        dw = torch.zeros(3, 3, K, N, dtype=torch.float32, device="cuda")
        L.conv3x3_wgrad_mfma_bf16(xb.data_ptr(), dyb.data_ptr(), dw.data_ptr(), ws.data_ptr(), wsb, b, H, W, K, N, S())
        torch.cuda.synchronize()
        return dw
    full = wgrad(x, dy)
    halves = wgrad(x[:B // 2].contiguous(), dy[:B // 2].contiguous()) + wgrad(x[B // 2:].contiguous(), dy[B // 2:].contiguous())
    close(host(halves), host(full), 2e-5, "batch additivity")
    ref = torch.zeros(3, 3, K, N, dtype=torch.float32, device="cuda")
    L.conv2d_direct_wgrad(x.data_ptr(), BF16, dy.data_ptr(), BF16, ref.data_ptr(), None, B, H, W, K, N, 3, S())
    torch.cuda.synchronize()
    close(host(full), host(ref), 1e-4, "MFMA filter gradient vs the fp32 direct kernel")


def test_wgrad_deferred_multi_layer_reduction(L):
    """phx_conv3x3_wgrad_mfma_bf16_partial + ONE phx_wgrad_reduce_multi over several layers == the per-layer launches."""
    import ctypes
    shapes = [(3, 16, 16, 64, 64), (2, 32, 32, 32, 96), (5, 8, 8, 96, 32), (2, 16, 48, 128, 64), (64, 2, 2, 64, 64)]
    keep, jobs, want, blk = [], [], [], 0
    for (B, H, W, K, N) in shapes:
        x, dy = dev(RNG.standard_normal((B, H, W, K)), BF16), dev(RNG.standard_normal((B, H, W, N)), BF16)
        wsb = int(L.conv3x3_wgrad_ws_bytes(B, H, W, K, N))
        ws = torch.empty(max(wsb // 4, 1), dtype=torch.float32).cuda()
        ref = torch.full((3, 3, K, N), 0.25, dtype=torch.float32).cuda()          # accumulate semantics: start from non-zero
        L.conv3x3_wgrad_mfma_bf16(x.data_ptr(), dy.data_ptr(), ref.data_ptr(), ws.data_ptr(), wsb, B, H, W, K, N, S())
        torch.cuda.synchronize()
        ws2 = torch.empty_like(ws)
        dw = torch.full((3, 3, K, N), 0.25, dtype=torch.float32).cuda()
        plan = (ctypes.c_int * 6)()
        L.conv3x3_wgrad_reduce_plan(B, H, W, K, N, plan)
        L.conv3x3_wgrad_mfma_bf16_partial(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws2.data_ptr(), wsb, B, H, W, K, N, S())
        if plan[0]:
            jobs.append((ws2.data_ptr(), dw.data_ptr(), plan[1], K, N, plan[2], plan[3], plan[4], plan[5], blk))
            blk += plan[4] * plan[5]
        keep.append((x, dy, ws, ws2)); want.append((ref, dw, plan[0]))
    assert any(w[2] for w in want) and not all(w[2] for w in want)                 # both the workspace and the atomics path
    rec = np.zeros(len(jobs), dtype=[("ws", "<u8"), ("dw", "<u8"), ("nslice", "<i4"), ("cin", "<i4"), ("cout", "<i4"), ("tci", "<i4"),
                                     ("tco", "<i4"), ("gx", "<i4"), ("gy", "<i4"), ("blk0", "<i4")])
    for i, j in enumerate(jobs):
        rec[i] = j
    desc = torch.from_numpy(rec.view(np.uint8).copy()).cuda()
    L.wgrad_reduce_multi(desc.data_ptr(), len(jobs), blk, S())
    torch.cuda.synchronize()
    for k, (ref, dw, _) in enumerate(want):
        close(host(dw), host(ref), 2e-6, "deferred reduction, layer %d" % k)


@pytest.mark.parametrize("case", [(64, 128, 128, 32), (64, 128, 128, 128), (64, 64, 64, 192), (64, 32, 32, 128)])
def test_batch_norm_full_size_vs_fp64_reference(L, case):
    """The engine's batch-norm launches at BASELINE sizes (statistics, fused apply + ReLU, replicated backward reduction, fused
    backward apply), where the CPU oracle takes minutes: against the same formulas (tfwrapper/normalisation.py:145-163 and their
    gradient) evaluated in float64 on the device from the same bf16 tensors.  Elements whose pre-activation lies within 1e-4 of the
    ReLU kink are left out of the dx comparison (fp32 and fp64 may put them on different sides)."""
    B, H, W, C = case
    P, eps, nrep = B * H * W, 1e-3, 4
    g = torch.Generator(device="cuda").manual_seed(5)
    x = (torch.randn(P, C, device="cuda", generator=g) * 1.5 + 0.3).to(torch.bfloat16)
    dA = torch.randn(P, C, device="cuda", generator=g).to(torch.bfloat16)
    gamma = (1.0 + 0.2 * torch.randn(C, device="cuda", generator=g)).float()
    beta = (0.1 * torch.randn(C, device="cuda", generator=g)).float()
    x64, d64, g64, b64 = x.double(), dA.double(), gamma.double(), beta.double()
    mean_r = x64.mean(0)
    var_r = ((x64 - mean_r) ** 2).mean(0)
    rstd_r = 1.0 / torch.sqrt(var_r + eps)
    xhat = (x64 - mean_r) * rstd_r
    pre = g64 * xhat + b64
    a_ref = torch.relu(pre)
    gq = d64 * (pre > 0)
    dbeta_r, dgamma_r = gq.sum(0), (gq * xhat).sum(0)
    dx_r = g64 * rstd_r * (gq - dbeta_r / P - xhat * dgamma_r / P)
    sums = torch.zeros(C, 2, dtype=torch.float32, device="cuda")
    pivot = torch.zeros(C, dtype=torch.float32, device="cuda")
    L.norm_stats(x.data_ptr(), BF16, sums.data_ptr(), pivot.data_ptr(), 1, P, C, S())
    a = torch.empty_like(x)
    mean, rstd, scale, shift = (torch.empty(C, dtype=torch.float32, device="cuda") for _ in range(4))
    L.norm_apply_fused(x.data_ptr(), BF16, sums.data_ptr(), pivot.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps, a.data_ptr(), BF16,
                       mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), None, None, 0.0, 1, P, C, C, 1, S())
    torch.cuda.synchronize()
    # (fp32 accumulation of 10^6 shifted values per channel: partial sums of magnitude 10^6 round at 0.1, 1.8e-5 measured)
    close(host(mean), mean_r.cpu().numpy(), 5e-5, "mean")
    close(host(rstd), rstd_r.cpu().numpy(), 5e-5, "rstd")
    assert float((a.double() - a_ref).abs().max()) <= 6e-3 * float(a_ref.abs().max())
    sums2 = torch.zeros(nrep, C, 2, dtype=torch.float32, device="cuda")
    L.norm_bwd_reduce(dA.data_ptr(), BF16, x.data_ptr(), BF16, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                      sums2.data_ptr(), 1, P, C, C, 1, nrep, S())
    dx = torch.empty_like(x)
    dgamma, dbeta = torch.zeros(C, dtype=torch.float32, device="cuda"), torch.zeros(C, dtype=torch.float32, device="cuda")
    L.norm_bwd_apply_fused(dA.data_ptr(), BF16, x.data_ptr(), BF16, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                           gamma.data_ptr(), sums2.data_ptr(), dx.data_ptr(), BF16, dgamma.data_ptr(), dbeta.data_ptr(), 1, P, C, C, 1,
                           nrep, S())
    torch.cuda.synchronize()
    close(host(dbeta), dbeta_r.cpu().numpy(), 1e-3, "dbeta")
    close(host(dgamma), dgamma_r.cpu().numpy(), 1e-3, "dgamma")
    away = pre.abs() > 1e-4
    assert float(away.double().mean()) > 0.999
    assert float(((dx.double() - dx_r) * away).abs().max()) <= 8e-3 * float(dx_r.abs().max())


@pytest.mark.parametrize("case", [(64, 32, 32, 128), (64, 16, 16, 192), (64, 16, 16, 384), (64, 8, 8, 192), (64, 32, 32, 64), (7, 12, 12, 96),
                                  (64, 32, 32, 32), (3, 5, 7, 8), (64, 16, 16, 64)])
def test_bn_bwd_onepass_vs_float64_and_the_two_pass_kernels(L, case):
    """phx_bn_bwd_onepass (round 6: batch-norm backward in ONE launch, (dA, x) held in registers across a grid barrier) at the sizes the
    benchmark runs it and on ragged ones: dx, dgamma, dbeta against the defining formulas in float64 on the device (as the two-pass
    test above) AND against phx_norm_bwd_reduce + phx_norm_bwd_apply_fused on the same tensors -- dx within one bf16 rounding step of
    the two-pass result (the per-channel sums differ in summation order only).  The barrier must not have timed out, and a second
    launch on re-zeroed state gives the same answer with other launches in flight beside it."""
    B, H, W, C = case
    P, eps, nrep = B * H * W, 1e-3, 4
    if not L.bn_bwd_onepass_supported(P, C, 1):
        pytest.skip("tensor too large for the register-resident form")
    g = torch.Generator(device="cuda").manual_seed(11)
    x = (torch.randn(P, C, device="cuda", generator=g) * 1.5 + 0.3).to(torch.bfloat16)
    dA = torch.randn(P, C, device="cuda", generator=g).to(torch.bfloat16)
    gamma = (1.0 + 0.2 * torch.randn(C, device="cuda", generator=g)).float()
    beta = (0.1 * torch.randn(C, device="cuda", generator=g)).float()
    x64, d64, g64, b64 = x.double(), dA.double(), gamma.double(), beta.double()
    mean_r = x64.mean(0)
    rstd_r = 1.0 / torch.sqrt(((x64 - mean_r) ** 2).mean(0) + eps)
    xhat = (x64 - mean_r) * rstd_r
    pre = g64 * xhat + b64
    gq = d64 * (pre > 0)
    dbeta_r, dgamma_r = gq.sum(0), (gq * xhat).sum(0)
    dx_r = g64 * rstd_r * (gq - dbeta_r / P - xhat * dgamma_r / P)
    mean, rstd = mean_r.float().contiguous(), rstd_r.float().contiguous()
    scale = (gamma * rstd).contiguous()
    shift = (beta - mean * scale).contiguous()
    # two-pass reference launches
    sums2 = torch.zeros(nrep, C, 2, dtype=torch.float32, device="cuda")
    L.norm_bwd_reduce(dA.data_ptr(), BF16, x.data_ptr(), BF16, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                      sums2.data_ptr(), 1, P, C, C, 1, nrep, S())
    dx2 = torch.empty_like(x)
    dg2, db2 = torch.zeros(C, dtype=torch.float32, device="cuda"), torch.zeros(C, dtype=torch.float32, device="cuda")
    L.norm_bwd_apply_fused(dA.data_ptr(), BF16, x.data_ptr(), BF16, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                           gamma.data_ptr(), sums2.data_ptr(), dx2.data_ptr(), BF16, dg2.data_ptr(), db2.data_ptr(), 1, P, C, C, 1, nrep, S())
    nbar = int(L.bn_bwd_onepass_barrier_words())
    outs = []
    side = torch.empty(1 << 24, dtype=torch.float32, device="cuda")
    s2 = torch.cuda.Stream()
    for rep in range(2):
        sums1 = torch.zeros(nrep, C, 2, dtype=torch.float32, device="cuda")
        bar = torch.zeros(nbar, dtype=torch.int32, device="cuda")
        dx1 = torch.full_like(x, float("nan"))
        dg1, db1 = torch.zeros(C, dtype=torch.float32, device="cuda"), torch.zeros(C, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        if rep == 1:                        # other work in flight on another stream while the barrier kernel runs
            with torch.cuda.stream(s2):
                for _ in range(4):
                    side.mul_(1.0001)
        L.bn_bwd_onepass(dA.data_ptr(), x.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                         sums1.data_ptr(), bar.data_ptr(), dx1.data_ptr(), dg1.data_ptr(), db1.data_ptr(), P, C, 1, nrep, S())
        torch.cuda.synchronize()
        assert int(bar[288]) == 0, "grid barrier timed out"
        outs.append((dx1.double(), host(dg1), host(db1)))
    for dx1, dg1, db1 in outs:
        close(db1, dbeta_r.cpu().numpy(), 1e-3, "dbeta")
        close(dg1, dgamma_r.cpu().numpy(), 1e-3, "dgamma")
        close(db1, host(db2), 2e-5, "dbeta vs two-pass")
        close(dg1, host(dg2), 2e-5, "dgamma vs two-pass")
        away = pre.abs() > 1e-4
        assert float(((dx1 - dx_r) * away).abs().max()) <= 8e-3 * float(dx_r.abs().max())
        # against the two-pass launches: the same formula on sums that differ in the last bits -> at most one bf16 step apart
        d = (dx1 - dx2.double()).abs()
        assert float(d.max()) <= 2.0 ** -7 * float(dx_r.abs().max()), float(d.max())
        assert float((d > 0).double().mean()) < 0.02


NORM_CASES = [
    # kind, B, H, W, C, G, dt
    ("batch", 3, 8, 8, 32, None, F32),
    ("batch", 2, 16, 16, 6, None, F32),
    ("batch", 2, 16, 16, 192, None, BF16),
    ("group", 3, 8, 8, 32, 2, F32),
    ("group", 2, 4, 4, 24, 2, F32),
    ("group", 2, 8, 8, 192, 12, BF16),
    ("instance", 3, 8, 8, 12, None, F32),
]


@pytest.mark.parametrize("case", NORM_CASES)
def test_norm_fwd_bwd(L, case):
    kind, B, H, W, C, G, dt = case
    x = RNG.standard_normal((B, H, W, C)) * 1.5 + 0.3
    gamma = 1.0 + 0.2 * RNG.standard_normal(C)
    beta = 0.1 * RNG.standard_normal(C)
    xr = rounded(x, dt).requires_grad_(True)
    gr = torch.as_tensor(gamma, dtype=torch.float32).double().requires_grad_(True)
    br = torch.as_tensor(beta, dtype=torch.float32).double().requires_grad_(True)
    if kind == "batch":
        NS, P, GG, eps = 1, B * H * W, C, 1e-3
        yr, mean_r, varu_r = T.batch_norm_train(xr, gr, br)
    elif kind == "group":
        NS, P, GG, eps = B, H * W, G, 1e-5
        yr = T.group_norm(xr, gr, br, G)
    else:
        NS, P, GG, eps = B, H * W, C, 1e-5
        yr = T.instance_norm(xr, gr, br)
    ar = T.relu(yr)
    xd, gd, bd = dev(x, dt), dev(gamma), dev(beta)
    sums = torch.zeros(NS, C, 2, dtype=torch.float32).cuda()
    pivot = torch.zeros(NS, C, dtype=torch.float32).cuda()
    L.norm_stats(xd.data_ptr(), dt, sums.data_ptr(), pivot.data_ptr(), NS, P, C, S())
    mean = torch.empty(NS * GG, dtype=torch.float32).cuda()
    rstd = torch.empty_like(mean)
    scale = torch.empty(NS * C, dtype=torch.float32).cuda()
    shift = torch.empty_like(scale)
    mm = dev(0.1 * RNG.standard_normal(C))
    mv = dev(1.0 + 0.3 * RNG.random(C))
    mm0, mv0 = host(mm).copy(), host(mv).copy()
    L.norm_finalize(sums.data_ptr(), pivot.data_ptr(), gd.data_ptr(), bd.data_ptr(), eps, NS, P, C, GG, mean.data_ptr(),
                    rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                    mm.data_ptr() if kind == "batch" else None, mv.data_ptr() if kind == "batch" else None,
                    0.01 if kind == "batch" else 0.0, S())
    a = torch.empty(B, H, W, C, dtype=tdt(dt)).cuda()
    L.affine_act(xd.data_ptr(), dt, scale.data_ptr(), shift.data_ptr(), a.data_ptr(), dt, NS, P, C, 1, S())
    close(host(a), ar.detach().numpy(), 2e-5 if dt == F32 else 6e-3, kind + " fwd")
    if kind == "batch":
        close(host(mm), mm0 - (mm0 - mean_r.detach().numpy()) * 0.01, 1e-5, "moving_mean")
        close(host(mv), mv0 - (mv0 - varu_r.detach().numpy()) * 0.01, 1e-5, "moving_var")
    dA = RNG.standard_normal((B, H, W, C))
    dAr = rounded(dA, dt)
    (ar * dAr).sum().backward()
    dAd = dev(dA, dt)
    sums2 = torch.zeros(NS, C, 2, dtype=torch.float32).cuda()
    L.norm_bwd_reduce(dAd.data_ptr(), dt, xd.data_ptr(), dt, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                      rstd.data_ptr(), sums2.data_ptr(), NS, P, C, GG, 1, 1, S())
    Sg = torch.empty(NS * GG * 2, dtype=torch.float32).cuda()
    dgamma = torch.zeros(C, dtype=torch.float32).cuda()
    dbeta = torch.zeros(C, dtype=torch.float32).cuda()
    L.norm_bwd_finalize(sums2.data_ptr(), gd.data_ptr(), Sg.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), NS, C,
                        GG, S())
    dx = torch.empty(B, H, W, C, dtype=tdt(dt)).cuda()
    L.norm_bwd_apply(dAd.data_ptr(), dt, xd.data_ptr(), dt, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                     rstd.data_ptr(), gd.data_ptr(), Sg.data_ptr(), dx.data_ptr(), dt, NS, P, C, GG, 1, S())
    close(host(dx), xr.grad.numpy(), 5e-5 if dt == F32 else 8e-3, kind + " dx")
    close(host(dgamma), gr.grad.numpy(), 5e-5 if dt == F32 else 2e-3, kind + " dgamma")
    close(host(dbeta), br.grad.numpy(), 5e-5 if dt == F32 else 2e-3, kind + " dbeta")
    # fused variants (the ones the engine emits): statistics finalised inside the apply kernels, replicated bwd accumulators
    a2 = torch.empty(B, H, W, C, dtype=tdt(dt)).cuda()
    mean2, rstd2, scale2, shift2 = (torch.empty_like(t) for t in (mean, rstd, scale, shift))
    L.norm_apply_fused(xd.data_ptr(), dt, sums.data_ptr(), pivot.data_ptr(), gd.data_ptr(), bd.data_ptr(), eps, a2.data_ptr(), dt,
                       mean2.data_ptr(), rstd2.data_ptr(), scale2.data_ptr(), shift2.data_ptr(), None, None, 0.0, NS, P, C, GG,
                       1, S())
    close(host(a2), ar.detach().numpy(), 2e-5 if dt == F32 else 6e-3, kind + " fused fwd")
    close(host(scale2), host(scale), 1e-6, kind + " fused scale")
    nrep = 3
    sums2r = torch.zeros(nrep, NS, C, 2, dtype=torch.float32).cuda()
    L.norm_bwd_reduce(dAd.data_ptr(), dt, xd.data_ptr(), dt, scale2.data_ptr(), shift2.data_ptr(), mean2.data_ptr(),
                      rstd2.data_ptr(), sums2r.data_ptr(), NS, P, C, GG, 1, nrep, S())
    close(host(sums2r).sum(axis=0), host(sums2), 1e-5 if dt == F32 else 1e-4, kind + " replicated sums2")
    dx2 = torch.empty(B, H, W, C, dtype=tdt(dt)).cuda()
    dgamma2 = torch.zeros(C, dtype=torch.float32).cuda()
    dbeta2 = torch.zeros(C, dtype=torch.float32).cuda()
    L.norm_bwd_apply_fused(dAd.data_ptr(), dt, xd.data_ptr(), dt, scale2.data_ptr(), shift2.data_ptr(), mean2.data_ptr(),
                           rstd2.data_ptr(), gd.data_ptr(), sums2r.data_ptr(), dx2.data_ptr(), dt, dgamma2.data_ptr(),
                           dbeta2.data_ptr(), NS, P, C, GG, 1, nrep, S())
    close(host(dx2), xr.grad.numpy(), 5e-5 if dt == F32 else 8e-3, kind + " fused dx")
    close(host(dgamma2), gr.grad.numpy(), 5e-5 if dt == F32 else 2e-3, kind + " fused dgamma")
    close(host(dbeta2), br.grad.numpy(), 5e-5 if dt == F32 else 2e-3, kind + " fused dbeta")
    # ... and with the gradient of a convolution bias in front of the normalisation (group / instance norm layers): the
    # per-channel sum of dx in closed form from the sums of both passes (phx_norm_bwd_apply_fused_bias), with and without pivot
    ref_db = xr.grad.numpy().sum(axis=(0, 1, 2))
    for pv in (pivot, None):
        fs = sums
        if pv is None:                                   # unshifted forward sums {sum x, sum x^2}
            fs = torch.zeros(NS, C, 2, dtype=torch.float32).cuda()
            L.norm_stats(xd.data_ptr(), dt, fs.data_ptr(), None, NS, P, C, S())
        dx3 = torch.empty(B, H, W, C, dtype=tdt(dt)).cuda()
        dg3, dbe3, dbias = (torch.zeros(C, dtype=torch.float32).cuda() for _ in range(3))
        L.norm_bwd_apply_fused_bias(dAd.data_ptr(), dt, xd.data_ptr(), dt, scale2.data_ptr(), shift2.data_ptr(), mean2.data_ptr(),
                                    rstd2.data_ptr(), gd.data_ptr(), sums2r.data_ptr(), dx3.data_ptr(), dt, dg3.data_ptr(),
                                    dbe3.data_ptr(), fs.data_ptr(), pv.data_ptr() if pv is not None else None, dbias.data_ptr(),
                                    NS, P, C, GG, 1, nrep, S())
        assert torch.equal(dx3, dx2)
        scale_db = max(1.0, float(np.abs(xr.grad.numpy()).sum(axis=(0, 1, 2)).max()))
        err = float(np.abs(host(dbias) - ref_db).max()) / scale_db
        assert err < (2e-5 if dt == F32 else 2e-3), (kind, "bias gradient", err)


@pytest.mark.parametrize("case", [(64, 2, 2, 192, 1), (64, 4, 4, 192, 1), (37, 3, 3, 48, 1), (64, 8, 8, 192, 1), (50, 8, 8, 16, 0),
                                  (1, 1, 2, 32, 1), (23, 6, 6, 128, 1)])
def test_bn_small_one_launch_layer(L, case):
    """phx_bn_small_fwd / _bwd (the H <= 8 levels: whole batch-norm layer in one launch) against the oracle's batch norm
    + ReLU and its autograd, bf16 activations; P from 2 to 4096 covers every slice depth (NIT 1, 2, 4, 8) and ragged tails."""
    B, H, W, C, act = case
    P = B * H * W
    assert L.bn_small_supported(P, C, BF16) == 1 and L.bn_small_supported(4097, C, BF16) == 0 and L.bn_small_supported(P, 24, BF16) == 0
    x = RNG.standard_normal((B, H, W, C)) * 1.5 + 0.3
    gamma = 1.0 + 0.2 * RNG.standard_normal(C)
    beta = 0.1 * RNG.standard_normal(C)
    xr = rounded(x, BF16).requires_grad_(True)
    gr = torch.as_tensor(gamma, dtype=torch.float32).double().requires_grad_(True)
    br = torch.as_tensor(beta, dtype=torch.float32).double().requires_grad_(True)
    yr, mean_r, varu_r = T.batch_norm_train(xr, gr, br)
    ar = T.relu(yr) if act else yr
    xd, gd, bd = dev(x, BF16), dev(gamma), dev(beta)
    mean, rstd, scale, shift = (torch.empty(C, dtype=torch.float32).cuda() for _ in range(4))
    mm = dev(0.1 * RNG.standard_normal(C))
    mv = dev(1.0 + 0.3 * RNG.random(C))
    mm0, mv0 = host(mm).copy(), host(mv).copy()
    a = torch.empty(B, H, W, C, dtype=torch.bfloat16).cuda()
    L.bn_small_fwd(xd.data_ptr(), BF16, gd.data_ptr(), bd.data_ptr(), 1e-3, a.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                   scale.data_ptr(), shift.data_ptr(), mm.data_ptr(), mv.data_ptr(), 0.01, P, C, act, S())
    close(host(a), ar.detach().numpy(), 6e-3, "bn_small fwd")
    close(host(mean), mean_r.detach().numpy(), 1e-5, "bn_small mean")
    if P > 1:
        close(host(mm), mm0 - (mm0 - mean_r.detach().numpy()) * 0.01, 1e-5, "bn_small moving_mean")
        close(host(mv), mv0 - (mv0 - varu_r.detach().numpy()) * 0.01, 1e-5, "bn_small moving_var")
    var_b = xr.detach().reshape(P, C).var(0, unbiased=False).numpy()
    close(host(rstd), 1.0 / np.sqrt(var_b + 1e-3), 1e-5, "bn_small rstd")
    close(host(scale), gamma.astype(np.float32) * host(rstd), 1e-6, "bn_small scale")
    # the three-launch path the engine uses on larger maps publishes the same statistics
    sums = torch.zeros(1, C, 2, dtype=torch.float32).cuda()
    pivot = torch.zeros(1, C, dtype=torch.float32).cuda()
    L.norm_stats(xd.data_ptr(), BF16, sums.data_ptr(), pivot.data_ptr(), 1, P, C, S())
    a3 = torch.empty_like(a)
    mean3, rstd3, scale3, shift3 = (torch.empty_like(t) for t in (mean, rstd, scale, shift))
    L.norm_apply_fused(xd.data_ptr(), BF16, sums.data_ptr(), pivot.data_ptr(), gd.data_ptr(), bd.data_ptr(), 1e-3, a3.data_ptr(),
                       BF16, mean3.data_ptr(), rstd3.data_ptr(), scale3.data_ptr(), shift3.data_ptr(), None, None, 0.0, 1, P, C, C,
                       act, S())
    close(host(scale), host(scale3), 2e-5, "bn_small vs three-launch scale")
    close(host(shift), host(shift3), 2e-5, "bn_small vs three-launch shift")
    dA = RNG.standard_normal((B, H, W, C))
    dAr = rounded(dA, BF16)
    (ar * dAr).sum().backward()
    dAd = dev(dA, BF16)
    dx = torch.empty(B, H, W, C, dtype=torch.bfloat16).cuda()
    dgamma = torch.full((C,), 0.5, dtype=torch.float32).cuda()        # accumulated (+=), not overwritten
    dbeta = torch.full((C,), -0.25, dtype=torch.float32).cuda()
    L.bn_small_bwd(dAd.data_ptr(), xd.data_ptr(), BF16, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                   gd.data_ptr(), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), P, C, act, S())
    close(host(dx), xr.grad.numpy(), 8e-3, "bn_small dx")
    close(host(dgamma) - 0.5, gr.grad.numpy(), 2e-3, "bn_small dgamma")
    close(host(dbeta) + 0.25, br.grad.numpy(), 2e-3, "bn_small dbeta")
    if P > 1024:
        return
    # fp32 pre-normalisation tensor (round 5: what phx_conv3x3_mfma_bf16_f32out leaves on the 2 x 2 / 4 x 4 levels).  The case bf16
    # storage cannot do: channels whose values spread by 2 % around a mean of 8 -- a bf16 grid step there (2^-4 = 0.0625) is 40 % of the
    # standard deviation, the fp32 path normalises the values as they are: compared with the oracle on the UNROUNDED input
    # (identity activation here: with a ReLU the handful of elements whose pre-activation is within fp32 round-off of zero take
    # the other branch than the float64 oracle, and one such element moves dbeta by a whole dA)
    xw = 8.0 + 0.16 * RNG.standard_normal((B, H, W, C))
    xw32 = torch.as_tensor(xw, dtype=torch.float32)
    xwr = xw32.double().requires_grad_(True)
    gr2, br2 = gr.detach().clone().requires_grad_(True), br.detach().clone().requires_grad_(True)
    aw, mean_w, _ = T.batch_norm_train(xwr, gr2, br2)
    xwd = xw32.cuda()
    a5 = torch.empty_like(a)
    L.bn_small_fwd(xwd.data_ptr(), F32, gd.data_ptr(), bd.data_ptr(), 1e-3, a5.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                   scale.data_ptr(), shift.data_ptr(), None, None, 0.0, P, C, 0, S())
    close(host(a5), aw.detach().numpy(), 6e-3, "bn_small fwd, fp32 input")
    close(host(mean), mean_w.detach().numpy(), 1e-5, "bn_small mean, fp32 input")
    if P >= 64:                          # the same layer through bf16 storage of x is off by tenths of the output's spread
        a6 = torch.empty_like(a)
        m6, r6, s6, h6 = (torch.empty_like(t) for t in (mean, rstd, scale, shift))
        L.bn_small_fwd(xwd.to(torch.bfloat16).data_ptr(), BF16, gd.data_ptr(), bd.data_ptr(), 1e-3, a6.data_ptr(), m6.data_ptr(), r6.data_ptr(),
                       s6.data_ptr(), h6.data_ptr(), None, None, 0.0, P, C, 0, S())
        e_bf16 = np.abs(host(a6) - aw.detach().numpy()).max()
        e_f32 = np.abs(host(a5) - aw.detach().numpy()).max()
        assert e_bf16 > 10 * e_f32 and e_bf16 > 0.1, (e_bf16, e_f32)
    (aw * dAr).sum().backward()
    dgamma.fill_(0.5); dbeta.fill_(-0.25)
    L.bn_small_bwd(dAd.data_ptr(), xwd.data_ptr(), F32, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                   gd.data_ptr(), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), P, C, 0, S())
    close(host(dx), xwr.grad.numpy(), 8e-3, "bn_small dx, fp32 input")            # (bf16 on output)
    close(host(dgamma) - 0.5, gr2.grad.numpy(), 2e-3, "bn_small dgamma, fp32 input")
    close(host(dbeta) + 0.25, br2.grad.numpy(), 2e-3, "bn_small dbeta, fp32 input")


@pytest.mark.parametrize("case", [(64, 2, 2, 192, 1, 6), (64, 4, 4, 192, 1, 3), (12, 4, 4, 64, 1, 1), (37, 3, 3, 36, 0, 5), (1, 1, 2, 32, 1, 2), (64, 4, 4, 32, 1, 9),
                                  (12, 8, 8, 192, 1, 4)])
def test_bn_wide_one_launch_layer_from_split_k_slices(L, case):
    """phx_bn_wide_fwd / _bwd (round 5; the 2 x 2 / 4 x 4 batch-norm layers: normalisation.py:145-163 + the activation of
    layers.py:134-135 in ONE launch that is also the split-K finishing pass of the convolution in front of it): the fp32 tensor given
    as nz slices, against the oracle's batch norm + ReLU and its autograd on the UNROUNDED sum; the backward pass from a bf16 dA
    and from the slices a split-K data gradient leaves.  Also against phx_bn_small_fwd / _bwd on the summed tensor (same math, other
    blocking)."""
    B, H, W, C, act, nz = case
    P = B * H * W
    assert L.bn_wide_supported(P, C) == 1 and L.bn_wide_supported(1025, C) == 0 and L.bn_wide_supported(P, 30) == 0
    x = (3.0 + 0.4 * RNG.standard_normal((B, H, W, C))).astype(np.float32)
    parts = RNG.standard_normal((nz, B, H, W, C)).astype(np.float32)
    parts[nz - 1] = x - parts[:nz - 1].sum(axis=0)
    pd = torch.as_tensor(parts).cuda()
    xs = pd[0].clone()
    for z in range(1, nz):
        xs += pd[z]                                          # slice order: what the kernel (and k_splitk_finish) computes
    gamma = 1.0 + 0.2 * RNG.standard_normal(C)
    beta = 0.1 * RNG.standard_normal(C)
    xr = xs.double().cpu().requires_grad_(True)
    gr = torch.as_tensor(gamma, dtype=torch.float32).double().requires_grad_(True)
    br = torch.as_tensor(beta, dtype=torch.float32).double().requires_grad_(True)
    yr, mean_r, varu_r = T.batch_norm_train(xr, gr, br)
    ar = T.relu(yr) if act else yr
    gd, bd = dev(gamma), dev(beta)
    mean, rstd, scale, shift = (torch.empty(C, dtype=torch.float32).cuda() for _ in range(4))
    mm, mv = dev(0.1 * RNG.standard_normal(C)), dev(1.0 + 0.3 * RNG.random(C))
    mm0, mv0 = host(mm).copy(), host(mv).copy()
    a = torch.empty(B, H, W, C, dtype=torch.bfloat16).cuda()
    xsum = torch.full((B, H, W, C), 5.0, dtype=torch.float32).cuda()
    L.bn_wide_fwd(pd.data_ptr(), nz, xsum.data_ptr(), gd.data_ptr(), bd.data_ptr(), 1e-3, a.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                  scale.data_ptr(), shift.data_ptr(), mm.data_ptr(), mv.data_ptr(), 0.01, P, C, act, S())
    if nz > 1:
        assert torch.equal(xsum, xs)                         # the finishing pass's sum, bit for bit
    close(host(a), ar.detach().numpy(), 6e-3, "bn_wide fwd")
    close(host(mean), mean_r.detach().numpy(), 1e-5, "bn_wide mean")
    if P > 1:
        close(host(mm), mm0 - (mm0 - mean_r.detach().numpy()) * 0.01, 1e-5, "bn_wide moving_mean")
        close(host(mv), mv0 - (mv0 - varu_r.detach().numpy()) * 0.01, 1e-5, "bn_wide moving_var")
    if C % 16 == 0:                                          # same layer through the 16-channel blocking
        a2 = torch.empty_like(a)
        m2, r2, s2, h2 = (torch.empty_like(t) for t in (mean, rstd, scale, shift))
        L.bn_small_fwd(xs.data_ptr(), F32, gd.data_ptr(), bd.data_ptr(), 1e-3, a2.data_ptr(), m2.data_ptr(), r2.data_ptr(), s2.data_ptr(),
                       h2.data_ptr(), None, None, 0.0, P, C, act, S())
        close(host(scale), host(s2), 2e-5, "bn_wide vs bn_small scale")
        close(host(shift), host(h2), 2e-5, "bn_wide vs bn_small shift")
    dA = RNG.standard_normal((B, H, W, C))
    dAr = rounded(dA, BF16)
    (ar * dAr).sum().backward()
    dAd = dev(dA, BF16)
    dx = torch.empty(B, H, W, C, dtype=torch.bfloat16).cuda()
    dgamma = torch.full((C,), 0.5, dtype=torch.float32).cuda()        # accumulated (+=), not overwritten
    dbeta = torch.full((C,), -0.25, dtype=torch.float32).cuda()
    L.bn_wide_bwd(dAd.data_ptr(), None, 0, xs.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gd.data_ptr(),
                  dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), P, C, act, S())
    ref = xr.grad.numpy()
    flips = 0
    if act:                                                  # (elements whose pre-activation is within fp32 round-off of zero may take the other ReLU branch)
        flips = int((np.abs(yr.detach().numpy()) < 1e-5).sum())
    if not flips:
        close(host(dx), ref, 8e-3, "bn_wide dx")
        close(host(dgamma) - 0.5, gr.grad.numpy(), 2e-3, "bn_wide dgamma")
        close(host(dbeta) + 0.25, br.grad.numpy(), 2e-3, "bn_wide dbeta")
    # dA as the fp32 slices of a split-K data gradient: summed, rounded to bf16 as the finishing pass would, same result bit for bit
    nzd = max(2, nz)
    dparts = RNG.standard_normal((nzd, B, H, W, C)).astype(np.float32)
    dpd = torch.as_tensor(dparts).cuda()
    dsum = dpd[0].clone()
    for z in range(1, nzd):
        dsum += dpd[z]
    dA2 = dsum.to(torch.bfloat16)
    dx_a, dx_b = torch.empty_like(dx), torch.empty_like(dx)
    dg_a, db_a, dg_b, db_b = (torch.zeros(C, dtype=torch.float32).cuda() for _ in range(4))
    L.bn_wide_bwd(dA2.data_ptr(), None, 0, xs.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gd.data_ptr(),
                  dx_a.data_ptr(), dg_a.data_ptr(), db_a.data_ptr(), P, C, act, S())
    L.bn_wide_bwd(None, dpd.data_ptr(), nzd, xs.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gd.data_ptr(),
                  dx_b.data_ptr(), dg_b.data_ptr(), db_b.data_ptr(), P, C, act, S())
    assert torch.equal(dx_a, dx_b) and torch.equal(dg_a, dg_b) and torch.equal(db_a, db_b)


@pytest.mark.parametrize("case", [("batch", 3, 16, 16, 32, 1), ("batch", 5, 8, 12, 192, 1), ("batch", 2, 64, 32, 64, 0), ("group", 3, 16, 8, 64, 1),
                                  ("instance", 2, 6, 10, 48, 1), ("batch", 64, 4, 4, 192, 1), ("batch", 9, 128, 128, 32, 1)])
def test_norm_apply_with_the_average_pool_fused_in(L, case):
    """phx_norm_apply_pool (round 5): the apply pass of a layer whose output also feeds averagepool2D (tfwrapper/layers.py:44-54;
    posteriors.py:80-82, priors.py:76-78) writes the 2 x 2 averages too -- bit-identical to phx_norm_apply_fused followed by
    phx_avgpool2x2_fwd (same statistics finalisation, the average taken of the values as stored, in the pool kernel's order), batch /
    group / instance norm, rectangular maps; and both against the oracle."""
    kind, B, H, W, C, act = case
    x = RNG.standard_normal((B, H, W, C)) * 1.5 + 0.3
    gamma, beta = 1.0 + 0.2 * RNG.standard_normal(C), 0.1 * RNG.standard_normal(C)
    xd, gd, bd = dev(x, BF16), dev(gamma), dev(beta)
    NS, P, G = (1, B * H * W, C) if kind == "batch" else (B, H * W, C if kind == "instance" else max(2, C // 16))
    eps = 1e-3 if kind == "batch" else 1e-5
    sums = torch.zeros(NS, C, 2, dtype=torch.float32).cuda()
    pivot = torch.zeros(NS, C, dtype=torch.float32).cuda()
    L.norm_stats(xd.data_ptr(), BF16, sums.data_ptr(), pivot.data_ptr(), NS, P, C, S())
    assert L.norm_apply_pool_supported(H, W, C) == 1 and L.norm_apply_pool_supported(H + 1, W, C) == 0

    def bufs():
        return [torch.empty(NS * G).cuda(), torch.empty(NS * G).cuda(), torch.empty(NS * C).cuda(), torch.empty(NS * C).cuda()]
    a1, a2 = torch.empty_like(xd), torch.empty_like(xd)
    p1, p2 = (torch.full((B, H // 2, W // 2, C), 9.0, dtype=torch.bfloat16).cuda() for _ in range(2))
    st1, st2 = bufs(), bufs()
    mm1, mv1 = dev(0.1 * RNG.standard_normal(C)), dev(1.0 + 0.3 * RNG.random(C))
    mm2, mv2 = mm1.clone(), mv1.clone()
    mom = 0.01 if kind == "batch" else 0.0
    L.norm_apply_fused(xd.data_ptr(), BF16, sums.data_ptr(), pivot.data_ptr(), gd.data_ptr(), bd.data_ptr(), eps, a1.data_ptr(), BF16,
                       *[t.data_ptr() for t in st1], mm1.data_ptr() if mom else None, mv1.data_ptr() if mom else None, mom, NS, P, C, G, act, S())
    L.avgpool2x2_fwd(a1.data_ptr(), BF16, p1.data_ptr(), B, H, W, C, S())
    L.norm_apply_pool(xd.data_ptr(), sums.data_ptr(), pivot.data_ptr(), gd.data_ptr(), bd.data_ptr(), eps, a2.data_ptr(), p2.data_ptr(),
                      *[t.data_ptr() for t in st2], mm2.data_ptr() if mom else None, mv2.data_ptr() if mom else None, mom, NS, P, C, G, H, W, act, S())
    torch.cuda.synchronize()
    assert torch.equal(a1, a2) and torch.equal(p1, p2)
    for u, v in zip(st1 + [mm1, mv1], st2 + [mm2, mv2]):
        assert torch.equal(u, v)
    if B * H * W * C <= 1 << 21:
        xr = rounded(x, BF16)
        g64, b64 = torch.as_tensor(gamma, dtype=torch.float32).double(), torch.as_tensor(beta, dtype=torch.float32).double()
        yr = T.batch_norm_train(xr, g64, b64)[0] if kind == "batch" else (T.group_norm(xr, g64, b64, G) if kind == "group" else T.instance_norm(xr, g64, b64))
        ar = T.relu(yr) if act else yr
        close(host(a2), ar.numpy(), 6e-3, "apply + pool: a")
        close(host(p2), T.avg_pool_2x2_same(rounded(ar.float().numpy(), BF16)).numpy(), 6e-3, "apply + pool: pooled")


@pytest.mark.parametrize("case", [("group", 3, 8, 8, 32, 2, 1), ("group", 2, 4, 4, 192, 12, 1), ("group", 64, 2, 2, 192, 12, 1),
                                  ("group", 5, 16, 16, 64, 4, 1), ("group", 70, 4, 4, 32, 2, 0), ("group", 3, 12, 12, 48, 3, 1),
                                  ("group", 131, 2, 2, 32, 2, 1), ("group", 9, 5, 6, 32, 2, 1), ("group", 2, 11, 13, 32, 2, 1),
                                  ("instance", 3, 8, 8, 48, 48, 1), ("instance", 37, 3, 3, 16, 16, 1), ("instance", 2, 16, 16, 16, 16, 1)])
def test_norm_small_one_launch_layer(L, case):
    """phx_norm_small_fwd / _bwd (group norm with 16-channel groups / instance norm, maps of up to 256 pixels: the whole layer in one
    launch, a wave per (sample, 16-channel slice)) against the oracle's group_norm / instance_norm + ReLU and its autograd; also fed by split-K
    slices plus the convolution bias, and the closed-form bias gradient against the per-channel sum of the oracle's dx."""
    kind, B, H, W, C, G, act = case
    P = H * W
    assert L.norm_small_supported(B, P, C, G, BF16) == 1
    assert L.norm_small_supported(B, 257, C, G, BF16) == 0 and L.norm_small_supported(B, P, 64, 8, BF16) == 0
    x = RNG.standard_normal((B, H, W, C)) * 1.5 + 0.3
    gamma = 1.0 + 0.2 * RNG.standard_normal(C)
    beta = 0.1 * RNG.standard_normal(C)
    xr = rounded(x, BF16).requires_grad_(True)
    gr = torch.as_tensor(gamma, dtype=torch.float32).double().requires_grad_(True)
    br = torch.as_tensor(beta, dtype=torch.float32).double().requires_grad_(True)
    fn = (lambda t: T.group_norm(t, gr, br, G)) if kind == "group" else (lambda t: T.instance_norm(t, gr, br))
    yr = fn(xr)
    ar = T.relu(yr) if act else yr
    xd, gd, bd = dev(x, BF16), dev(gamma), dev(beta)
    mean = torch.empty(B * G, dtype=torch.float32).cuda()
    rstd = torch.empty_like(mean)
    scale = torch.empty(B * C, dtype=torch.float32).cuda()
    shift = torch.empty_like(scale)
    a = torch.empty(B, H, W, C, dtype=torch.bfloat16).cuda()
    L.norm_small_fwd(xd.data_ptr(), None, 0, None, gd.data_ptr(), bd.data_ptr(), 1e-5, a.data_ptr(), mean.data_ptr(),
                     rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), B, P, C, G, act, S())
    close(host(a), ar.detach().numpy(), 6e-3, "norm_small fwd")
    xg = xr.detach().reshape(B, P, G, C // G).permute(0, 2, 1, 3).reshape(B, G, -1)
    close(host(mean).reshape(B, G), xg.mean(-1).numpy(), 1e-5, "norm_small mean")
    close(host(rstd).reshape(B, G), 1.0 / np.sqrt(xg.var(-1, unbiased=False).numpy() + 1e-5), 2e-5, "norm_small rstd")
    # the three-launch path used on larger maps publishes the same statistics
    sums = torch.zeros(B, C, 2, dtype=torch.float32).cuda()
    pivot = torch.zeros(B, C, dtype=torch.float32).cuda()
    L.norm_stats(xd.data_ptr(), BF16, sums.data_ptr(), pivot.data_ptr(), B, P, C, S())
    a3 = torch.empty_like(a)
    mean3, rstd3, scale3, shift3 = (torch.empty_like(t) for t in (mean, rstd, scale, shift))
    L.norm_apply_fused(xd.data_ptr(), BF16, sums.data_ptr(), pivot.data_ptr(), gd.data_ptr(), bd.data_ptr(), 1e-5, a3.data_ptr(),
                       BF16, mean3.data_ptr(), rstd3.data_ptr(), scale3.data_ptr(), shift3.data_ptr(), None, None, 0.0, B, P, C, G,
                       act, S())
    close(host(scale), host(scale3), 5e-5, "norm_small vs three-launch scale")
    close(host(shift), host(shift3), 5e-5, "norm_small vs three-launch shift")
    # fed by split-K slices + the convolution bias
    nz = 3
    cb = RNG.standard_normal(C).astype(np.float32) * 0.3
    parts = RNG.standard_normal((nz, B, H, W, C)).astype(np.float32)
    parts[nz - 1] = x.astype(np.float32) - cb - parts[:nz - 1].sum(axis=0)
    pd, cbd = torch.as_tensor(parts).cuda(), torch.as_tensor(cb).cuda()
    xs = pd.sum(0) + cbd
    xsr = xs.to(torch.bfloat16).double().cpu()
    ys = (lambda t: T.group_norm(t, gr.detach(), br.detach(), G) if kind == "group" else T.instance_norm(t, gr.detach(), br.detach()))(xsr)
    a4, x4 = torch.empty_like(a), torch.empty_like(a)
    L.norm_small_fwd(x4.data_ptr(), pd.data_ptr(), nz, cbd.data_ptr(), gd.data_ptr(), bd.data_ptr(), 1e-5, a4.data_ptr(),
                     mean3.data_ptr(), rstd3.data_ptr(), scale3.data_ptr(), shift3.data_ptr(), B, P, C, G, act, S())
    close(host(x4), xsr.numpy(), 4e-3, "norm_small split-K summed input")
    close(host(a4), (T.relu(ys) if act else ys).numpy(), 8e-3, "norm_small split-K fwd")
    dA = RNG.standard_normal((B, H, W, C))
    dAr = rounded(dA, BF16)
    (ar * dAr).sum().backward()
    dAd = dev(dA, BF16)
    dx = torch.empty(B, H, W, C, dtype=torch.bfloat16).cuda()
    dgamma = torch.full((C,), 0.5, dtype=torch.float32).cuda()        # accumulated (+=), not overwritten
    dbeta = torch.full((C,), -0.25, dtype=torch.float32).cuda()
    dbias = torch.full((C,), 2.0, dtype=torch.float32).cuda()
    L.norm_small_bwd(dAd.data_ptr(), xd.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                     gd.data_ptr(), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), dbias.data_ptr(), B, P, C, G, act, S())
    close(host(dx), xr.grad.numpy(), 8e-3, "norm_small dx")
    close(host(dgamma) - 0.5, gr.grad.numpy(), 2e-3, "norm_small dgamma")
    close(host(dbeta) + 0.25, br.grad.numpy(), 2e-3, "norm_small dbeta")
    ref_db = xr.grad.numpy().sum(axis=(0, 1, 2))
    scale_db = max(1.0, float(np.abs(xr.grad.numpy()).sum(axis=(0, 1, 2)).max()))
    got = host(dbias) - 2.0
    if kind == "instance":
        assert np.all(got == 0.0)                      # per-channel statistics: identically zero, not computed
        assert float(np.abs(ref_db).max()) / scale_db < 1e-6
    else:
        assert float(np.abs(got - ref_db).max()) / scale_db < 2e-3, ("norm_small bias gradient", got[:4], ref_db[:4])
    # dbias == NULL: same dx
    dx2 = torch.empty_like(dx)
    L.norm_small_bwd(dAd.data_ptr(), xd.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                     gd.data_ptr(), dx2.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), None, B, P, C, G, act, S())
    assert torch.equal(dx, dx2)


def test_bn_infer_scale_shift(L):
    C = 24
    g, b, mm, mv = RNG.random(C) + 0.5, RNG.standard_normal(C), RNG.standard_normal(C), RNG.random(C) + 0.5
    x = RNG.standard_normal((2, 4, 4, C))
    sc, sh = torch.empty(C).cuda(), torch.empty(C).cuda()
    gd, bd, mmd, mvd, xd = dev(g), dev(b), dev(mm), dev(mv), dev(x)      # keep the device buffers alive
    L.bn_infer_scale_shift(gd.data_ptr(), bd.data_ptr(), mmd.data_ptr(), mvd.data_ptr(), 1e-3, C,
                           sc.data_ptr(), sh.data_ptr(), S())
    y = torch.empty(2, 4, 4, C).cuda()
    L.affine_act(xd.data_ptr(), F32, sc.data_ptr(), sh.data_ptr(), y.data_ptr(), F32, 1, 32, C, 0, S())
    f = lambda v: torch.as_tensor(v, dtype=torch.float32).double()
    ref = T.batch_norm_infer(f(x), f(g), f(b), f(mm), f(mv))
    close(host(y), ref.numpy(), 1e-5)


@pytest.mark.parametrize("case", [(2, 8, 8, 16, F32), (2, 5, 3, 6, F32), (1, 6, 6, 32, BF16), (3, 2, 2, 24, F32)])
def test_avgpool_and_bilinear(L, case):
    B, H, W, C, dt = case
    x = RNG.standard_normal((B, H, W, C))
    xr = rounded(x, dt).requires_grad_(True)
    xd = dev(x, dt)
    tol = 1e-6 if dt == F32 else 6e-3
    # avg pool
    pr = T.avg_pool_2x2_same(xr)
    OH, OW = (H + 1) // 2, (W + 1) // 2
    p = torch.empty(B, OH, OW, C, dtype=tdt(dt)).cuda()
    L.avgpool2x2_fwd(xd.data_ptr(), dt, p.data_ptr(), B, H, W, C, S())
    close(host(p), pr.detach().numpy(), tol, "pool fwd")
    dp = RNG.standard_normal((B, OH, OW, C))
    (pr * rounded(dp, dt)).sum().backward()
    dx = torch.empty(B, H, W, C, dtype=tdt(dt)).cuda()
    dpd = dev(dp, dt)
    L.avgpool2x2_bwd(dpd.data_ptr(), dt, dx.data_ptr(), B, H, W, C, S())
    close(host(dx), xr.grad.numpy(), tol, "pool bwd")
    # bilinear x2 (TF1 legacy)
    xr.grad = None
    ur = T.resize_bilinear_legacy(xr, 2 * H, 2 * W)
    u = torch.empty(B, 2 * H, 2 * W, C, dtype=tdt(dt)).cuda()
    L.bilinear_up2x_fwd(xd.data_ptr(), dt, u.data_ptr(), B, H, W, C, S())
    close(host(u), ur.detach().numpy(), tol, "bilinear fwd")
    du = RNG.standard_normal((B, 2 * H, 2 * W, C))
    (ur * rounded(du, dt)).sum().backward()
    dud = dev(du, dt)
    L.bilinear_up2x_bwd(dud.data_ptr(), dt, dx.data_ptr(), B, H, W, C, S())
    close(host(dx), xr.grad.numpy(), tol if dt == F32 else 1e-2, "bilinear bwd")


@pytest.mark.parametrize("case", [(64, 64, 64, 64), (64, 128, 128, 32), (64, 32, 32, 192)])
def test_avgpool_and_bilinear_full_size(L, case):
    """BASELINE-size pooling / up-sampling launches (bf16): 2 x 2 average pooling against torch's on the device (the windows never
    straddle the border at even sizes, where TF's SAME padding and count-excluding average coincide with the plain one); its backward
    and the legacy bilinear x2 pair through the adjoint identity <op(x), g> = <x, op^T(g)> and, for the bilinear forward, the
    interpolation property that a constant image stays constant (each output is a convex combination)."""
    B, H, W, C = case
    g = torch.Generator(device="cuda").manual_seed(17)
    x = torch.randn(B, H, W, C, device="cuda", generator=g).to(torch.bfloat16)
    p = torch.empty(B, H // 2, W // 2, C, dtype=torch.bfloat16, device="cuda")
    L.avgpool2x2_fwd(x.data_ptr(), BF16, p.data_ptr(), B, H, W, C, S())
    ref = torch.nn.functional.avg_pool2d(x.float().permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    torch.cuda.synchronize()
    assert float((p.float() - ref).abs().max()) <= 6e-3 * float(ref.abs().max())
    gp = torch.randn(B, H // 2, W // 2, C, device="cuda", generator=g).to(torch.bfloat16)
    dx = torch.empty_like(x)
    L.avgpool2x2_bwd(gp.data_ptr(), BF16, dx.data_ptr(), B, H, W, C, S())
    u = torch.empty(B, 2 * H, 2 * W, C, dtype=torch.bfloat16, device="cuda")
    L.bilinear_up2x_fwd(x.data_ptr(), BF16, u.data_ptr(), B, H, W, C, S())
    gu = torch.randn(B, 2 * H, 2 * W, C, device="cuda", generator=g).to(torch.bfloat16)
    dxu = torch.empty_like(x)
    L.bilinear_up2x_bwd(gu.data_ptr(), BF16, dxu.data_ptr(), B, H, W, C, S())
    ones = torch.full_like(x, 1.5)
    uc = torch.empty_like(u)
    L.bilinear_up2x_fwd(ones.data_ptr(), BF16, uc.data_ptr(), B, H, W, C, S())
    torch.cuda.synchronize()
    assert float((uc.float() - 1.5).abs().max()) == 0.0

    def adjoint(out, gout, xin, gin, what):
        a, b = float((out.double() * gout.double()).sum()), float((xin.double() * gin.double()).sum())
        tol = 8e-3 * float(np.sqrt(float((out.double() ** 2).sum()) * float((gout.double() ** 2).sum()))) / np.sqrt(min(out.numel(), xin.numel()))
        assert abs(a - b) <= tol, (what, a, b, tol)
    adjoint(p, gp, x, dx, "average pool")
    adjoint(u, gu, x, dxu, "bilinear x2")


def test_concat_split_add_cast_misc(L):
    for (Ca, Cb, dt) in [(32, 32, BF16), (128, 64, BF16), (32, 6, F32), (1, 2, F32), (8, 4, F32)]:
        a, b = RNG.standard_normal((2, 4, 4, Ca)), RNG.standard_normal((2, 4, 4, Cb))
        ad, bd = dev(a, dt), dev(b, dt)
        out = torch.empty(2, 4, 4, Ca + Cb, dtype=tdt(dt)).cuda()
        L.concat2(ad.data_ptr(), Ca, bd.data_ptr(), Cb, out.data_ptr(), 32, dt, S())
        ref = torch.cat([ad, bd], dim=3)
        assert torch.equal(out, ref)
        a2, b2 = torch.empty_like(ad), torch.empty_like(bd)
        L.split2(out.data_ptr(), a2.data_ptr(), Ca, b2.data_ptr(), Cb, 32, dt, S())
        assert torch.equal(a2, ad) and torch.equal(b2, bd)
        L.add_inplace(a2.data_ptr(), ad.data_ptr(), a2.numel(), dt, S())
        close(host(a2), 2 * host(ad), 1e-2 if dt == BF16 else 1e-7)
        cs = torch.zeros(Ca, dtype=torch.float32).cuda()
        L.channel_sum_accumulate(ad.data_ptr(), dt, cs.data_ptr(), 32, Ca, S())
        close(host(cs), host(ad).sum(axis=(0, 1, 2)), 1e-5, "channel_sum")
    # the vectorised add (n >= 4096, multiple of 8): full trips of four vectors, a ragged tail, a grid capped at 4096 blocks
    for (n, dt) in [(4096, BF16), (8 * 1000 + 8, F32), (8 * (4 * 256 * 5 + 37), BF16), (8 * 256 * 4 * 4100 + 8 * 3, BF16), (4100, F32)]:
        a, b = torch.randn(n, device="cuda").to(tdt(dt)), torch.randn(n, device="cuda").to(tdt(dt))
        want = (a.float() + b.float()).to(tdt(dt))
        L.add_inplace(a.data_ptr(), b.data_ptr(), n, dt, S())
        assert torch.equal(a, want), (n, dt)
    for (npix, C, dt) in [(64, 32, BF16), (1000, 192, BF16), (5000, 64, F32), (70000, 32, BF16), (333, 8, F32)]:
        a = RNG.standard_normal((npix, C))           # the 16-byte-load kernel (bias gradients of the group / instance norm nets)
        ad = dev(a, dt)
        cs = torch.full((C,), 0.5, dtype=torch.float32).cuda()
        L.channel_sum_accumulate(ad.data_ptr(), dt, cs.data_ptr(), npix, C, S())
        close(host(cs), 0.5 + host(ad).astype(np.float64).sum(axis=0), 2e-5, "channel_sum v8 %s" % ((npix, C, dt),))
    x = RNG.standard_normal((2, 8, 8, 1)).astype(np.float32)
    s = RNG.integers(0, 4, (2, 8, 8)).astype(np.uint8)
    out = torch.empty(2, 8, 8, 5).cuda()
    xd, sdv = dev(x), torch.as_tensor(s).cuda()
    L.posterior_input(xd.data_ptr(), sdv.data_ptr(), out.data_ptr(), F32, 128, 4, S())
    ref = np.concatenate([x, np.eye(4)[s] - 0.5], axis=-1)
    close(host(out), ref, 1e-7)
    z = RNG.standard_normal((3, 6))
    bo = torch.empty(3, 10, 6).cuda()
    zd = dev(z)
    L.broadcast_pixels_fwd(zd.data_ptr(), bo.data_ptr(), F32, 3, 10, 6, S())
    close(host(bo), np.repeat(z[:, None, :], 10, 1), 1e-6)
    dz = torch.empty(3, 6).cuda()
    L.broadcast_pixels_bwd(bo.data_ptr(), F32, dz.data_ptr(), 3, 10, 6, S())
    close(host(dz), 10 * z, 1e-5)
    g = torch.empty(3, 6).cuda()
    L.global_avgpool_fwd(bo.data_ptr(), g.data_ptr(), 3, 10, 6, S())
    close(host(g), z, 1e-5)
    gb = torch.empty(3, 10, 6).cuda()
    L.global_avgpool_bwd(zd.data_ptr(), gb.data_ptr(), 3, 10, 6, S())
    close(host(gb), np.repeat(z[:, None, :], 10, 1) / 10, 1e-6)


@pytest.mark.parametrize("C,H,Ls", [(2, 64, 5), (4, 48, 5), (2, 32, 1), (3, 16, 3)])
def test_residual_ce(L, C, H, Ls):
    B = 3
    shifts = list(range(Ls))
    s_np = [RNG.standard_normal((B, H >> sh, H >> sh, C)) for sh in shifts]
    lab = RNG.integers(0, C, (B, H, H)).astype(np.uint8)
    sr = [torch.as_tensor(v, dtype=torch.float32).double().requires_grad_(True) for v in s_np]
    oh = T.one_hot(torch.as_tensor(lab), C, torch.float64)
    full = [T.resize_nearest(v, H, H) for v in sr]
    acc, losses_r, tot = None, [None] * Ls, 0.0
    for l in reversed(range(Ls)):
        acc = full[l] if acc is None else acc + full[l]
        losses_r[l] = T.multinoulli_loss_with_logits(oh, acc)
        tot = tot + 0.7 * losses_r[l]
    tot.backward()
    sd = [dev(v) for v in s_np]
    dsd = [torch.zeros_like(v) for v in sd]
    from phiseg_code_amd import runtime as rt
    losses = torch.zeros(8 + 512).cuda()
    s_out = torch.empty(B, H, H, C).cuda()
    sm = torch.empty(B, H, H, C).cuda()
    labd = torch.as_tensor(lab).cuda()
    L.residual_ce(rt.ptr_array([v.data_ptr() for v in sd]), rt.ptr_array([v.data_ptr() for v in dsd]),
                  rt.int_array(shifts), Ls, labd.data_ptr(), B, H, H, C, 0.7, 1.0 / B,
                  losses.data_ptr(), s_out.data_ptr(), sm.data_ptr(), S())
    close(host(losses)[:Ls], [float(v) for v in losses_r], 2e-5, "ce losses")
    close(host(s_out), acc.detach().numpy(), 1e-5, "s_out")
    close(host(sm), torch.softmax(acc, dim=-1).detach().numpy(), 1e-5, "softmax")
    for l in range(Ls):
        close(host(dsd[l]), sr[l].grad.numpy(), 3e-5, "ds[%d]" % l)


def test_kl_and_adam(L):
    n, B = 2 * 32 * 32 * 2, 2
    mu0, mu1 = RNG.standard_normal(n), RNG.standard_normal(n)
    s0, s1 = RNG.random(n) + 0.2, RNG.random(n) + 0.2
    f = lambda v: torch.as_tensor(v, dtype=torch.float32).double().reshape(B, -1).requires_grad_(True)
    a, b, c, d = f(mu0), f(s0), f(mu1), f(s1)
    kl = 16 * T.kl_two_gauss_with_diag_cov(a, b, c, d)
    (0.5 * kl).backward()
    loss = torch.zeros(1).cuda()
    outs = [torch.empty(n).cuda() for _ in range(4)]
    ins = [dev(mu0), dev(s0), dev(mu1), dev(s1)]
    L.kl_diag_gauss(ins[0].data_ptr(), ins[1].data_ptr(), ins[2].data_ptr(), ins[3].data_ptr(), n, 16.0,
                    1.0 / B, 0.5, loss.data_ptr(), *[o.data_ptr() for o in outs], S())
    close(host(loss), [float(kl)], 1e-5, "kl")
    for o, r in zip(outs, (a, b, c, d)):
        close(host(o), r.grad.reshape(-1).numpy(), 2e-5, "kl grad")
    # Adam, 3 steps, n not a multiple of 4
    n = 1003
    p, g = RNG.standard_normal(n), RNG.standard_normal(n)
    pd, gd = dev(p), dev(g)
    md, vd = torch.zeros(n).cuda(), torch.zeros(n).cuda()
    step = torch.zeros(1, dtype=torch.int32).cuda()
    lr = torch.tensor([1e-3]).cuda()
    pr = torch.as_tensor(p, dtype=torch.float32).double()
    gr = torch.as_tensor(g, dtype=torch.float32).double()
    mr, vr = torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
    for t in range(3):
        L.adam_tf1(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), n, lr.data_ptr(), 0.9, 0.999, 1e-8,
                   step.data_ptr(), S())
        L.step_increment(step.data_ptr(), S())
        pr, mr, vr = T.adam_tf1_step(pr, gr, mr, vr, t + 1, 1e-3)
    np.testing.assert_allclose(host(pd), pr.numpy(), rtol=0, atol=2e-6)
    assert int(step.cpu()[0]) == 3


def test_adam_full_parameter_arena(L):
    """phx_adam_tf1 over an arena of the benchmark's size (17.8 M parameters, not a multiple of 4) for three steps with a fresh
    gradient each: against the oracle's TF 1.12 epsilon-hat update evaluated in float64 on the device."""
    n = 17824003
    g = torch.Generator(device="cuda").manual_seed(19)
    p = torch.randn(n, device="cuda", generator=g)
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    lr = torch.tensor([1e-3], device="cuda")
    pr, mr, vr = p.double(), m.double(), v.double()
    for t in range(3):
        gr = torch.randn(n, device="cuda", generator=g) * (10.0 ** (t - 1))
        L.adam_tf1(p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr.data_ptr(), 0.9, 0.999, 1e-8, step.data_ptr(), S())
        L.step_increment(step.data_ptr(), S())
        pr, mr, vr = T.adam_tf1_step(pr, gr.double(), mr, vr, t + 1, 1e-3)
    torch.cuda.synchronize()
    assert float((p.double() - pr).abs().max()) <= 2e-6
    # (the slots are fp32 with fp32 coefficients, as TF's: 1 - 0.999f = 0.00100005, 4.7e-5 off the float64 oracle's 0.001)
    assert float((m.double() - mr).abs().max()) <= 1e-5 * float(mr.abs().max())
    assert float((v.double() - vr).abs().max()) <= 1e-4 * float(vr.abs().max())


def test_reparam_and_graph_capture(L):
    B, per = 4, 2 * 8 * 8
    mu, sg = RNG.standard_normal((B, per)), RNG.random((B, per)) + 0.1
    step = torch.tensor([2], dtype=torch.int32).cuda()
    z = torch.empty(B, per).cuda()
    mud, sgd = dev(mu), dev(sg)
    st = ctypes.c_void_p()
    L.stream_create(ctypes.byref(st))
    torch.cuda.synchronize()
    L.graph_begin_capture(st)
    L.reparam_fwd(mud.data_ptr(), sgd.data_ptr(), z.data_ptr(), B, per, 42, step.data_ptr(), 3, 0, st)
    L.step_increment(step.data_ptr(), st)
    ge = ctypes.c_void_p()
    L.graph_end_capture(st, ctypes.byref(ge))
    for it in range(2):          # the replay must pick up the incremented device-side step
        L.graph_launch(ge, st)
        L.stream_sync(st)
        eps = philox.normal(42, 2 + it, 3, B, per, dtype=np.float64)
        close(host(z), np.float32(mu).astype(np.float64) + np.float32(sg).astype(np.float64) * eps, 1e-5)
    dz = RNG.standard_normal((B, per))
    dsg = torch.empty(B, per).cuda()
    dzd = dev(dz)
    L.reparam_bwd(dzd.data_ptr(), dsg.data_ptr(), B, per, 42, step.data_ptr(), 3, 0, S())
    close(host(dsg), np.float32(dz).astype(np.float64) * philox.normal(42, 4, 3, B, per, dtype=np.float64), 1e-5)
    L.graph_destroy(ge)
    L.stream_destroy(st)


@pytest.mark.parametrize("case", [(3, 16, 16, 128, 2, BF16, "identity"), (2, 8, 8, 192, 2, BF16, "softplus"),
                                  (2, 4, 4, 32, 6, F32, "identity"), (2, 8, 8, 24, 4, F32, "softplus"),
                                  (1, 128, 128, 128, 2, BF16, "identity"), (2, 2, 2, 32, 8, BF16, "identity")])
def test_head1x1_kernels(L, case):
    B, H, W, C, NO, xdt, act = case
    npix = B * H * W
    x = RNG.standard_normal((B, H, W, C))
    w = RNG.standard_normal((1, 1, C, NO)) / np.sqrt(C)
    b = RNG.standard_normal(NO) * 0.3
    xr = rounded(x, xdt).requires_grad_(True)
    wr = torch.as_tensor(w, dtype=torch.float32).double().requires_grad_(True)
    br = torch.as_tensor(b, dtype=torch.float32).double().requires_grad_(True)
    yr = oracle_conv(xr, wr, br, act)
    xd, wd, bd = dev(x, xdt), dev(w), dev(b)
    y = torch.empty(B, H, W, NO, dtype=torch.float32).cuda()
    L.head1x1_fwd(xd.data_ptr(), xdt, wd.data_ptr(), bd.data_ptr(), y.data_ptr(), npix, C, NO, ACT[act], S())
    close(host(y), yr.detach().numpy(), 2e-5, "head fwd")
    dy = RNG.standard_normal((B, H, W, NO))
    pre = T.conv2d_same(xr, wr) + br.reshape(1, 1, 1, -1)
    (pre * torch.as_tensor(dy, dtype=torch.float32).double()).sum().backward()
    dyd = dev(dy)
    dx = torch.empty(B, H, W, C, dtype=tdt(xdt)).cuda()
    L.head1x1_dgrad(dyd.data_ptr(), wd.data_ptr(), dx.data_ptr(), xdt, npix, C, NO, S())
    close(host(dx), xr.grad.numpy(), 1e-5 if xdt == F32 else 6e-3, "head dgrad")
    dw = torch.zeros(C, NO, dtype=torch.float32).cuda()
    db = torch.zeros(NO, dtype=torch.float32).cuda()
    L.head1x1_wgrad(xd.data_ptr(), xdt, dyd.data_ptr(), dw.data_ptr(), db.data_ptr(), npix, C, NO, S())
    close(host(dw), wr.grad.numpy().reshape(C, NO), 3e-5, "head wgrad")
    close(host(db), br.grad.numpy(), 3e-5, "head dbias")


@pytest.mark.parametrize("case", [(64, 128, 128, 32, 2), (64, 64, 64, 64, 2), (16, 192, 192, 32, 4), (64, 16, 16, 192, 2)])
def test_head1x1_full_size_vs_device_matmul(L, case):
    """The 1x1 heads (mu / sigma / per-level logits: posteriors.py:125-127, likelihoods.py:220) at BASELINE sizes against float64
    matrix products on the device from the same bf16 activations: forward with bias, data gradient, filter / bias gradient."""
    B, H, W, C, NO = case
    npix = B * H * W
    g = torch.Generator(device="cuda").manual_seed(23)
    x = torch.randn(npix, C, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(C, NO, device="cuda", generator=g) / np.sqrt(C)).float()
    b = (torch.randn(NO, device="cuda", generator=g) * 0.3).float()
    dy = torch.randn(npix, NO, device="cuda", generator=g).float()
    y = torch.empty(npix, NO, dtype=torch.float32, device="cuda")
    L.head1x1_fwd(x.data_ptr(), BF16, w.data_ptr(), b.data_ptr(), y.data_ptr(), npix, C, NO, 0, S())
    dx = torch.empty(npix, C, dtype=torch.bfloat16, device="cuda")
    L.head1x1_dgrad(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), BF16, npix, C, NO, S())
    dw = torch.zeros(C, NO, dtype=torch.float32, device="cuda")
    db = torch.zeros(NO, dtype=torch.float32, device="cuda")
    L.head1x1_wgrad(x.data_ptr(), BF16, dy.data_ptr(), dw.data_ptr(), db.data_ptr(), npix, C, NO, S())
    torch.cuda.synchronize()
    x64, w64, d64 = x.double(), w.double(), dy.double()
    yr, dxr, dwr, dbr = x64 @ w64 + b.double(), d64 @ w64.t(), x64.t() @ d64, d64.sum(0)
    assert float((y.double() - yr).abs().max()) <= 2e-5 * float(yr.abs().max())
    assert float((dx.double() - dxr).abs().max()) <= 6e-3 * float(dxr.abs().max())
    # (sums over up to 10^6 pixels in fp32 partials: relative to the spread sqrt(npix) of such a sum)
    assert float((dw.double() - dwr).abs().max()) <= 1e-4 * np.sqrt(npix)
    assert float((db.double() - dbr).abs().max()) <= 1e-4 * np.sqrt(npix)


@pytest.mark.parametrize("nout", [2, 4])
def test_head1x1_wgrad_multi_matches_per_head_launches(L, nout):
    """One phx_head1x1_wgrad_multi launch over heads of different widths / map sizes == the per-head launches (accumulating).  Every
    second job is given the PRE-normalisation tensor and the layer's scale / shift / ReLU instead of a (round 5: a training plan does
    not write the activation whose only reader is a head): same result as the launch on the bf16 a that phx_affine_act materialises."""
    import ctypes
    heads = [(64, 2, 2, 192), (64, 16, 16, 192), (3, 32, 32, 128), (2, 64, 64, 64), (1, 128, 128, 32), (5, 7, 3, 40)]
    keep, rows, want, blk, lds = [], [], [], 0, 0
    for hi, (B, H, W, C) in enumerate(heads):
        npix = B * H * W
        ypre, dy = dev(RNG.standard_normal((npix, C)), BF16), dev(RNG.standard_normal((npix, nout)))
        xsc, xsh = dev(1.0 + 0.3 * RNG.standard_normal(C)), dev(0.2 * RNG.standard_normal(C))
        xform = hi % 2 == 1
        # a = bf16(relu(y * scale + shift)) with the library's own arithmetic (one fma in fp32: phx_affine_act)
        x = ypre
        if xform:
            x = torch.empty_like(ypre)
            L.affine_act(ypre.data_ptr(), BF16, xsc.data_ptr(), xsh.data_ptr(), x.data_ptr(), BF16, 1, npix, C, 1, S())
        dw_ref = torch.full((C, nout), 0.5, dtype=torch.float32).cuda()
        db_ref = torch.full((nout,), -1.0, dtype=torch.float32).cuda()
        L.head1x1_wgrad(x.data_ptr(), BF16, dy.data_ptr(), dw_ref.data_ptr(), db_ref.data_ptr(), npix, C, nout, S())
        dw = torch.full((C, nout), 0.5, dtype=torch.float32).cuda()
        db = torch.full((nout,), -1.0, dtype=torch.float32).cuda()
        plan = (ctypes.c_int * 4)()
        L.head1x1_wgrad_plan(npix, C, nout, plan)
        rows.append(((ypre if xform else x).data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), npix, C, plan[0], plan[1], blk,
                     xsc.data_ptr() if xform else 0, xsh.data_ptr() if xform else 0, 1 if xform else 0, 0))
        blk += plan[2]
        lds = max(lds, plan[3])
        keep.append((x, ypre, dy, xsc, xsh)); want.append((dw_ref, db_ref, dw, db, 1e-5))
    rec = np.zeros(len(rows), dtype=[("x", "<u8"), ("dy", "<u8"), ("dw", "<u8"), ("db", "<u8"), ("npix", "<u8"), ("C", "<i4"), ("PL", "<i4"),
                                     ("chunk", "<i4"), ("blk0", "<i4"), ("xscale", "<u8"), ("xshift", "<u8"), ("xact", "<i4"), ("pad", "<i4")])
    for i, r in enumerate(rows):
        rec[i] = r
    desc = torch.from_numpy(rec.view(np.uint8).copy()).cuda()
    L.head1x1_wgrad_multi(desc.data_ptr(), len(rows), blk, BF16, nout, lds, S())
    torch.cuda.synchronize()
    for k, (dw_ref, db_ref, dw, db, tol) in enumerate(want):
        close(host(dw), host(dw_ref), tol, "multi head wgrad, head %d" % k)
        close(host(db), host(db_ref), 1e-5, "multi head dbias, head %d" % k)


def test_wgrad_deferred_small_map_launches(L):
    """phx_conv3x3_wgrad_multi: the small-map filter gradients of several layers in one launch per kernel variant (+ the
    deferred reduction for those that use a workspace) == the per-layer phx_conv3x3_wgrad_mfma_bf16 launches."""
    import ctypes
    shapes = [(64, 2, 2, 64, 64), (64, 4, 4, 64, 64), (16, 8, 8, 64, 64), (64, 8, 8, 64, 64), (9, 4, 4, 32, 96), (64, 2, 2, 96, 32),
              (3, 8, 8, 32, 32), (2, 16, 16, 64, 64), (16, 16, 16, 64, 64), (40, 16, 16, 96, 64), (7, 16, 32, 32, 32)]
    nb = int(L.conv3x3_wgrad_multi_job_bytes())
    groups, rjobs, keep, want = {}, [], [], []
    for (B, H, W, K, N) in shapes:
        x, dy = dev(RNG.standard_normal((B, H, W, K)), BF16), dev(RNG.standard_normal((B, H, W, N)), BF16)
        wsb = int(L.conv3x3_wgrad_ws_bytes(B, H, W, K, N))
        ws, ws2 = (torch.empty(max(wsb // 4, 1), dtype=torch.float32).cuda() for _ in range(2))
        ref = torch.full((3, 3, K, N), 0.25, dtype=torch.float32).cuda()
        L.conv3x3_wgrad_mfma_bf16(x.data_ptr(), dy.data_ptr(), ref.data_ptr(), ws.data_ptr(), wsb, B, H, W, K, N, S())
        dw = torch.full((3, 3, K, N), 0.25, dtype=torch.float32).cuda()
        jb, info = ctypes.create_string_buffer(nb), (ctypes.c_int * 9)()
        tgt = (0, 24, 96)[len(keep) % 3]                     # this job's own pixel-tile split (0: the stand-alone plan)
        L.conv3x3_wgrad_multi_job(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws2.data_ptr(), wsb, B, H, W, K, N, tgt, 0, jb, info)
        keep.append((x, dy, ws, ws2))
        if not info[0]:                                       # a 16x16-tile shape with <= 4 tiles adds straight into dw: per-layer launch
            # (with the LDS-DMA kernels switched off -- PHX_WGRAD_DMA=0, a debug hook -- no 16x16-tile layer is deferred)
            assert (B, H, W) == (2, 16, 16)
            continue
        g = groups.setdefault(int(info[0]), dict(recs=[], blocks=0, lds=0))
        L.conv3x3_wgrad_multi_job(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws2.data_ptr(), wsb, B, H, W, K, N, tgt, g["blocks"], jb, info)
        g["recs"].append(jb.raw); g["blocks"] += info[1]; g["lds"] = max(g["lds"], info[2])
        if info[3]:
            rjobs.append((ws2.data_ptr(), dw.data_ptr(), info[4], K, N, info[5], info[6], info[7], info[8]))
        want.append((ref, dw))
    assert len(groups) >= 5 and any(v > 8 for v in groups) and rjobs         # register-staged and LDS-DMA variants
    for variant, g in groups.items():
        desc = torch.frombuffer(bytearray(b"".join(g["recs"])), dtype=torch.uint8).cuda()
        L.conv3x3_wgrad_multi(desc.data_ptr(), len(g["recs"]), g["blocks"], variant, g["lds"], S())
        keep.append(desc)
    rec = np.zeros(len(rjobs), dtype=[("ws", "<u8"), ("dw", "<u8"), ("nslice", "<i4"), ("cin", "<i4"), ("cout", "<i4"), ("tci", "<i4"),
                                      ("tco", "<i4"), ("gx", "<i4"), ("gy", "<i4"), ("blk0", "<i4")])
    blk = 0
    for i, j in enumerate(rjobs):
        rec[i] = j + (blk,)
        blk += j[7] * j[8]
    desc = torch.from_numpy(rec.view(np.uint8).copy()).cuda()
    L.wgrad_reduce_multi(desc.data_ptr(), len(rjobs), blk, S())
    torch.cuda.synchronize()
    for k, (ref, dw) in enumerate(want):
        close(host(dw), host(ref), 3e-6, "deferred small-map filter gradient, layer %d" % k)


def test_abi_rejects_bad_arguments_loudly(L):
    """Invalid shapes / arguments come back as PhxError with the library's message -- nothing is launched, nothing falls back."""
    from phiseg_code_amd.runtime import PhxError
    x = torch.zeros(2 * 16 * 16 * 64, dtype=torch.bfloat16).cuda()
    y = torch.zeros(2 * 16 * 16 * 64, dtype=torch.bfloat16).cuda()
    w = torch.zeros(9 * 64 * 64, dtype=torch.bfloat16).cuda()
    f = torch.zeros(9 * 64 * 64, dtype=torch.float32).cuda()
    with pytest.raises(PhxError, match="K % 32"):
        L.conv3x3_mfma_bf16(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, 0, None, 2, 16, 16, 48, 64, S())
    with pytest.raises(PhxError, match="y == NULL"):
        L.conv3x3_mfma_bf16(x.data_ptr(), w.data_ptr(), None, None, 0, None, 2, 16, 16, 64, 64, S())
    with pytest.raises(PhxError, match="workspace too small"):
        L.conv3x3_wgrad_mfma_bf16(x.data_ptr(), y.data_ptr(), f.data_ptr(), f.data_ptr(), 16, 2, 16, 16, 64, 64, S())
    with pytest.raises(PhxError, match="ksize"):
        L.conv2d_direct(x.data_ptr(), BF16, f.data_ptr(), None, y.data_ptr(), BF16, 2, 16, 16, 64, 64, 2, 0, 0, None, S())
    with pytest.raises(PhxError, match="bad shape"):
        L.conv2d_direct(x.data_ptr(), BF16, f.data_ptr(), None, y.data_ptr(), BF16, 0, 16, 16, 64, 64, 3, 0, 0, None, S())
    with pytest.raises(PhxError, match="nout"):
        L.head1x1_wgrad(x.data_ptr(), BF16, f.data_ptr(), f.data_ptr(), f.data_ptr(), 512, 64, 3, S())
    with pytest.raises(PhxError, match="P <= 4096"):
        L.bn_small_fwd(x.data_ptr(), BF16, f.data_ptr(), f.data_ptr(), 1e-3, y.data_ptr(), f.data_ptr(), f.data_ptr(), f.data_ptr(), f.data_ptr(),
                       None, None, 0.0, 5000, 64, 1, S())
    with pytest.raises(PhxError, match="empty job list"):
        L.wgrad_reduce_multi(None, 0, 0, S())
    torch.cuda.synchronize()


def test_comm_abi_single_rank_rccl(L):
    """phx_comm_* on RCCL with a world of one (the only world a 1-GPU box offers): rendezvous id, communicator, a bucketed
    in-place all-reduce enqueued on a HIP stream (identity for one rank), teardown; bad arguments are rejected."""
    import ctypes
    from phiseg_code_amd import runtime as rt
    idbuf = ctypes.create_string_buffer(128)
    L.comm_unique_id(idbuf)
    assert any(idbuf.raw)
    comm = ctypes.c_void_p()
    L.comm_init(ctypes.byref(comm), 1, 0, idbuf.raw)
    x = torch.randn(1 << 20, device="cuda")
    ref = x.clone()
    L.comm_allreduce_sum_f32(comm, x.data_ptr(), x.numel(), 1 << 18, S())
    torch.cuda.synchronize()
    assert torch.equal(x, ref)
    with pytest.raises(rt.PhxError):
        L.comm_allreduce_sum_f32(None, x.data_ptr(), x.numel(), 0, S())
    with pytest.raises(rt.PhxError):
        L.comm_init(ctypes.byref(ctypes.c_void_p()), 2, 5, idbuf.raw)
    L.comm_destroy(comm)


@pytest.mark.parametrize("case", [(2, 16, 16, 32, 64), (3, 4, 4, 192, 64), (1, 32, 64, 96, 128), (2, 12, 12, 32, 32), (40, 2, 2, 64, 96),
                                  (1, 16, 64, 32, 32), (2, 16, 32, 64, 96)])
@pytest.mark.parametrize("act", ["relu", "identity"])
def test_conv3x3_mfma_affine_epilogue(Ld, case, act):
    """Inference-mode batch norm + activation folded into the convolution (reference: conv2d -> batch_norm(is_training=False)
    -> relu, tfwrapper/layers.py:123-135, normalisation.py:145-163): y = act(conv(x) * scale + shift) in one launch, on the
    256-pixel tiles, the split-K small-map path and (forced) the large-map kernels."""
    L = Ld                                   # (the test build: this test sets the kernel policy)
    B, H, W, K, N = case
    x = RNG.standard_normal((B, H, W, K))
    w = RNG.standard_normal((3, 3, K, N)) / np.sqrt(9 * K)
    gamma, beta = 1.0 + 0.2 * RNG.standard_normal(N), 0.1 * RNG.standard_normal(N)
    mm, mv = 0.1 * RNG.standard_normal(N), 1.0 + 0.3 * RNG.random(N)
    eps = 1e-3
    xr, wr = rounded(x, BF16), rounded(w, BF16)
    pre = T.conv2d_same(xr, wr)
    sc = gamma / np.sqrt(mv + eps)
    ref = pre * torch.as_tensor(sc) + torch.as_tensor(beta - mm * sc)
    if act == "relu":
        ref = T.relu(ref)
    xd, wd = dev(x, BF16), dev(w)
    wf = torch.empty(9 * N * K, dtype=torch.bfloat16).cuda()
    wg = torch.empty(9 * N * K, dtype=torch.bfloat16).cuda()
    L.pack_conv3x3_bf16(wd.data_ptr(), wf.data_ptr(), wg.data_ptr(), K, N, S())
    vec = [dev(v) for v in (gamma, beta, mm, mv)]
    scale, shift = torch.empty(N, dtype=torch.float32).cuda(), torch.empty(N, dtype=torch.float32).cuda()
    desc = np.zeros(1, dtype=[("gamma", "<u8"), ("beta", "<u8"), ("mm", "<u8"), ("mv", "<u8"), ("scale", "<u8"), ("shift", "<u8"),
                              ("C", "<i4"), ("eps", "<f4")])
    desc[0] = (vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(), scale.data_ptr(), shift.data_ptr(), N, eps)
    dd = torch.from_numpy(desc.view(np.uint8).copy()).cuda()
    L.bn_infer_scale_shift_multi(dd.data_ptr(), 1, S())
    close(host(scale), sc, 1e-6, "scale")
    wsb = int(L.conv3x3_mfma_ws_bytes(B, H, W, K, N))
    ws = torch.empty(max(wsb // 4, 1), dtype=torch.float32).cuda()
    y = torch.empty(B, H, W, N, dtype=torch.bfloat16).cuda()
    for env in (1, 2, 0):
        L.debug_conv_policy(env, 1)
        try:
            y.zero_()
            L.conv3x3_mfma_bf16_affine(xd.data_ptr(), wf.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr(), ACT[act],
                                       ws.data_ptr() if wsb else None, wsb, B, H, W, K, N, S())
            close(host(y), ref.numpy(), 6e-3, "affine epilogue, large-map policy %d" % env)
        finally:
            L.debug_conv_policy(1, 1)


@pytest.mark.parametrize("case", [(3, 16, 16, 32, 2, BF16), (2, 8, 8, 192, 2, BF16), (5, 2, 2, 192, 2, BF16), (2, 4, 4, 64, 6, BF16),
                                  (1, 32, 32, 64, 4, F32)])
def test_latent_heads_fused(L, case):
    """phx_latent_heads_fwd / _bwd -- mu = x Wmu + bmu, sigma = softplus(x Wsig + bsig), z = mu + sigma * eps in one launch, and its
    backward in one launch (posteriors.py:125-128, priors.py:117-120) -- against the oracle (Philox noise contract included) and
    against the five launches they replace."""
    B, H, W, C, Z, dt = case
    P, hw = B * H * W, H * W
    x = RNG.standard_normal((P, C))
    wmu, wsg = RNG.standard_normal((C, Z)) / np.sqrt(C), RNG.standard_normal((C, Z)) / np.sqrt(C)
    bmu, bsg = RNG.standard_normal(Z) * 0.2, RNG.standard_normal(Z) * 0.2
    xd, wmud, wsgd, bmud, bsgd = dev(x, dt), dev(wmu), dev(wsg), dev(bmu), dev(bsg)
    step = torch.tensor([3], dtype=torch.int32).cuda()
    seed, sid, off = 42 + (1 << 33), 17, 5
    mu, sg, z = (torch.empty(P, Z, dtype=torch.float32).cuda() for _ in range(3))
    L.latent_heads_fwd(xd.data_ptr(), dt, wmud.data_ptr(), bmud.data_ptr(), wsgd.data_ptr(), bsgd.data_ptr(), mu.data_ptr(),
                       sg.data_ptr(), z.data_ptr(), P, C, Z, hw, seed, step.data_ptr(), sid, off, S())
    xr = rounded(x, dt).requires_grad_(True)
    wmur, wsgr = torch.as_tensor(wmu, dtype=torch.float32).double().requires_grad_(True), torch.as_tensor(wsg, dtype=torch.float32).double().requires_grad_(True)
    eps = torch.as_tensor(philox.normal(seed, 3, sid, B, hw * Z, sample_offset=off, dtype=np.float64)).reshape(P, Z)
    mur = xr @ wmur + torch.as_tensor(bmu, dtype=torch.float32).double()
    sgr = T.softplus(xr @ wsgr + torch.as_tensor(bsg, dtype=torch.float32).double())
    zr = mur + sgr * eps
    close(host(mu), mur.detach().numpy(), 2e-5, "mu")
    close(host(sg), sgr.detach().numpy(), 2e-5, "sigma")
    close(host(z), zr.detach().numpy(), 2e-5, "z")
    # the three launches it replaces
    mu2, sg2, z2 = (torch.empty(P, Z, dtype=torch.float32).cuda() for _ in range(3))
    L.head1x1_fwd(xd.data_ptr(), dt, wmud.data_ptr(), bmud.data_ptr(), mu2.data_ptr(), P, C, Z, 0, S())
    L.head1x1_fwd(xd.data_ptr(), dt, wsgd.data_ptr(), bsgd.data_ptr(), sg2.data_ptr(), P, C, Z, 2, S())
    L.reparam_fwd(mu2.data_ptr(), sg2.data_ptr(), z2.data_ptr(), B, hw * Z, seed, step.data_ptr(), sid, off, S())
    close(host(mu), host(mu2), 1e-6, "mu vs head kernel")
    close(host(z), host(z2), 1e-6, "z vs head + reparam kernels")
    # heads only (z == NULL: a prior whose sample is not consumed)
    mu3, sg3 = torch.empty_like(mu), torch.empty_like(sg)
    L.latent_heads_fwd(xd.data_ptr(), dt, wmud.data_ptr(), bmud.data_ptr(), wsgd.data_ptr(), bsgd.data_ptr(), mu3.data_ptr(),
                       sg3.data_ptr(), None, P, C, Z, hw, seed, None, 0, 0, S())
    assert torch.equal(mu3, mu) and torch.equal(sg3, sg)
    # backward: L = sum(z * gz) + sum(mu * gm) + sum(sigma * gs)
    gz, gm, gs = RNG.standard_normal((P, Z)), RNG.standard_normal((P, Z)), RNG.standard_normal((P, Z))
    ((zr * torch.as_tensor(gz)).sum() + (mur * torch.as_tensor(gm)).sum() + (sgr * torch.as_tensor(gs)).sum()).backward()
    gzd, gmd, gsd = dev(gz), dev(gm), dev(gs)
    dx = torch.empty(P, C, dtype=tdt(dt)).cuda()
    gmu, gsig = torch.empty(P, Z, dtype=torch.float32).cuda(), torch.empty(P, Z, dtype=torch.float32).cuda()
    L.latent_heads_bwd(gzd.data_ptr(), gmd.data_ptr(), gsd.data_ptr(), sg.data_ptr(), wmud.data_ptr(), wsgd.data_ptr(), dx.data_ptr(), dt,
                       gmu.data_ptr(), gsig.data_ptr(), P, C, Z, hw, seed, step.data_ptr(), sid, off, S())
    close(host(dx), xr.grad.numpy(), 2e-5 if dt == F32 else 6e-3, "dx")
    dwm, dws = torch.zeros(C, Z, dtype=torch.float32).cuda(), torch.zeros(C, Z, dtype=torch.float32).cuda()
    dbm, dbs = torch.zeros(Z, dtype=torch.float32).cuda(), torch.zeros(Z, dtype=torch.float32).cuda()
    L.head1x1_wgrad(xd.data_ptr(), dt, gmu.data_ptr(), dwm.data_ptr(), dbm.data_ptr(), P, C, Z, S())
    L.head1x1_wgrad(xd.data_ptr(), dt, gsig.data_ptr(), dws.data_ptr(), dbs.data_ptr(), P, C, Z, S())
    close(host(dwm), wmur.grad.numpy(), 5e-5, "dWmu")
    close(host(dws), wsgr.grad.numpy(), 5e-5, "dWsigma")
    # no upstream sample gradient (heads feeding the KL term only)
    dx2 = torch.empty_like(dx)
    L.latent_heads_bwd(None, gmd.data_ptr(), gsd.data_ptr(), sg.data_ptr(), wmud.data_ptr(), wsgd.data_ptr(), dx2.data_ptr(), dt,
                       gmu.data_ptr(), gsig.data_ptr(), P, C, Z, hw, seed, None, 0, 0, S())
    xr.grad = None
    xr2 = rounded(x, dt).requires_grad_(True)
    mur2 = xr2 @ wmur.detach() + torch.as_tensor(bmu, dtype=torch.float32).double()
    sgr2 = T.softplus(xr2 @ wsgr.detach() + torch.as_tensor(bsg, dtype=torch.float32).double())
    ((mur2 * torch.as_tensor(gm)).sum() + (sgr2 * torch.as_tensor(gs)).sum()).backward()
    close(host(dx2), xr2.grad.numpy(), 2e-5 if dt == F32 else 6e-3, "dx (KL only)")


@pytest.mark.parametrize("dt", [F32, BF16])
def test_accumulating_pool_and_resize_gradients(L, dt):
    """phx_avgpool2x2_bwd_acc / phx_bilinear_up2x_bwd_acc: the gradient of a tensor with several readers is accumulated in place by the
    later contributions (the engine then needs neither a second buffer nor an add pass): result = previous content + plain backward."""
    B, H, W, C = 3, 6, 10, 16
    prev = RNG.standard_normal((B, H, W, C))
    dyp = RNG.standard_normal((B, (H + 1) // 2, (W + 1) // 2, C))
    dyu = RNG.standard_normal((B, 2 * H, 2 * W, C))
    tol = 1e-6 if dt == F32 else 1.2e-2
    for fn, fn_acc, dy in ((L.avgpool2x2_bwd, L.avgpool2x2_bwd_acc, dyp), (L.bilinear_up2x_bwd, L.bilinear_up2x_bwd_acc, dyu)):
        dyd = dev(dy, dt)
        plain = torch.empty(B, H, W, C, dtype=tdt(dt)).cuda()
        fn(dyd.data_ptr(), dt, plain.data_ptr(), B, H, W, C, S())
        acc = dev(prev, dt)
        fn_acc(dyd.data_ptr(), dt, acc.data_ptr(), B, H, W, C, S())
        close(host(acc), host(plain) + rounded(prev, dt).numpy(), tol, fn_acc.__name__)


@pytest.mark.parametrize("case", [("batch", 2, 16, 16, 128, 2), ("batch", 3, 8, 24, 32, 4), ("group", 2, 16, 16, 64, 2)])
def test_norm_layer_with_fused_head(L, case):
    """phx_norm_apply_fused_head / phx_norm_bwd_reduce_head / phx_norm_bwd_apply_fused_head: a 1x1 head that is the only reader of
    a = relu(norm(y)) (the likelihood's top layer feeding y_lvl0, likelihoods.py:220) computed inside the apply pass, and in the
    backward pass dA = dy_head w_head^T formed on the fly -- against the separate launches they replace (phx_head1x1_fwd /
    phx_head1x1_dgrad + the plain norm kernels), which are themselves checked against the oracle above."""
    kind, B, H, W, C, NO = case
    NS = 1 if kind == "batch" else B
    G = C if kind == "batch" else 4
    P = B * H * W if kind == "batch" else H * W
    eps = 1e-3 if kind == "batch" else 1e-5
    npix = B * H * W
    y = dev(RNG.standard_normal((B, H, W, C)) * 1.5 + 0.3, BF16)
    gamma, beta = dev(1.0 + 0.2 * RNG.standard_normal(C)), dev(0.3 * RNG.standard_normal(C))
    wh, bh = dev(RNG.standard_normal((C, NO)) / np.sqrt(C)), dev(RNG.standard_normal(NO) * 0.2)
    yf = y.float()
    red = (0, 1, 2) if kind == "batch" else (1, 2)
    sums = torch.stack([yf.sum(dim=red), (yf * yf).sum(dim=red)], dim=-1).reshape(NS, C, 2).contiguous()
    assert L.norm_head_supported(C, NO, BF16, BF16) == 1 and L.norm_head_supported(192, 2, BF16, BF16) == 0

    def bufs():
        return (torch.empty(B, H, W, C, dtype=torch.bfloat16).cuda(), torch.empty(NS * G).cuda(), torch.empty(NS * G).cuda(),
                torch.empty(NS * C).cuda(), torch.empty(NS * C).cuda())
    a1, mean, rstd, scale, shift = bufs()
    L.norm_apply_fused(y.data_ptr(), BF16, sums.data_ptr(), None, gamma.data_ptr(), beta.data_ptr(), eps, a1.data_ptr(), BF16,
                           mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), None, None, 0.0, NS, P, C, G, 1, S())
    yh1 = torch.empty(npix, NO, dtype=torch.float32).cuda()
    L.head1x1_fwd(a1.data_ptr(), BF16, wh.data_ptr(), bh.data_ptr(), yh1.data_ptr(), npix, C, NO, 0, S())
    a2, mean2, rstd2, scale2, shift2 = bufs()
    yh2 = torch.empty(npix, NO, dtype=torch.float32).cuda()
    L.norm_apply_fused_head(y.data_ptr(), BF16, sums.data_ptr(), None, gamma.data_ptr(), beta.data_ptr(), eps, a2.data_ptr(), BF16,
                            mean2.data_ptr(), rstd2.data_ptr(), scale2.data_ptr(), shift2.data_ptr(), None, None, 0.0, NS, P, C, G, 1,
                            wh.data_ptr(), bh.data_ptr(), NO, yh2.data_ptr(), S())
    assert torch.equal(a1, a2) and torch.equal(scale, scale2)
    close(host(yh2), host(yh1), 2e-6, "head output")
    # y == NULL (training plans, round 5): a is not written, the head output is the same
    yh3 = torch.empty_like(yh2)
    L.norm_apply_fused_head(y.data_ptr(), BF16, sums.data_ptr(), None, gamma.data_ptr(), beta.data_ptr(), eps, None, BF16,
                            mean2.data_ptr(), rstd2.data_ptr(), scale2.data_ptr(), shift2.data_ptr(), None, None, 0.0, NS, P, C, G, 1,
                            wh.data_ptr(), bh.data_ptr(), NO, yh3.data_ptr(), S())
    assert torch.equal(yh3, yh2)
    # backward
    dyh = dev(RNG.standard_normal((npix, NO)))
    dA = torch.empty(B, H, W, C, dtype=torch.bfloat16).cuda()
    L.head1x1_dgrad(dyh.data_ptr(), wh.data_ptr(), dA.data_ptr(), BF16, npix, C, NO, S())
    nrep = 3
    s_ref = torch.zeros(nrep, NS, C, 2).cuda()
    L.norm_bwd_reduce(dA.data_ptr(), BF16, y.data_ptr(), BF16, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                      s_ref.data_ptr(), NS, P, C, G, 1, nrep, S())
    s_new = torch.zeros(nrep, NS, C, 2).cuda()
    L.norm_bwd_reduce_head(dyh.data_ptr(), wh.data_ptr(), NO, y.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                           rstd.data_ptr(), s_new.data_ptr(), NS, P, C, G, 1, nrep, S())
    close(host(s_new).sum(0), host(s_ref).sum(0), 1e-5, "backward sums")
    dx1, dx2 = torch.empty_like(dA), torch.empty_like(dA)
    dg1, db1, dg2, db2 = (torch.zeros(C).cuda() for _ in range(4))
    L.norm_bwd_apply_fused_bias(dA.data_ptr(), BF16, y.data_ptr(), BF16, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                rstd.data_ptr(), gamma.data_ptr(), s_ref.data_ptr(), dx1.data_ptr(), BF16, dg1.data_ptr(), db1.data_ptr(),
                                None, None, None, NS, P, C, G, 1, nrep, S())
    L.norm_bwd_apply_fused_head(dyh.data_ptr(), wh.data_ptr(), NO, y.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                rstd.data_ptr(), gamma.data_ptr(), s_ref.data_ptr(), dx2.data_ptr(), dg2.data_ptr(), db2.data_ptr(),
                                None, None, None, NS, P, C, G, 1, nrep, S())
    assert torch.equal(dx1, dx2)
    close(host(dg2), host(dg1), 1e-6, "dgamma")


@pytest.mark.parametrize("case", [(64, 4, 4, 192, 64, 192), (64, 8, 8, 192, 64, 192), (16, 16, 16, 192, 64, 192), (8, 16, 16, 192, 192, 192),
                                  (4, 32, 32, 128, 64, 128), (3, 32, 32, 128, 128, 192), (64, 2, 2, 192, 64, 96), (2, 16, 32, 32, 32, 128),
                                  (1, 32, 64, 64, 64, 192), (2, 16, 32, 64, 32, 96),
                                  # BASELINE-size concatenations (the default policy's kernels; the CPU oracle is skipped there)
                                  (64, 128, 128, 32, 32, 128), (64, 64, 64, 128, 64, 192)])
@pytest.mark.parametrize("force_dma", [0, 1, 2])
def test_conv3x3_concat_free(Ld, case, force_dma, policy):
    """Concat-free convolution (tf.concat([a, b], axis=3) -> conv2D 3x3: posteriors.py:87,120, priors.py:112, likelihoods.py:210):
    forward with a dual input, data gradient with a dual output and the filter gradient with a dual input equal the same launches on
    the materialised concatenation (forward / data gradient bit for bit), and the forward pass matches the oracle's concat + conv."""
    L = Ld                                   # (the test build: this test sets the kernel policy)
    from oracle import tf1_ops as O
    B, H, W, K1, K2, N = case
    if force_dma and B * H * W > 65536:
        pytest.skip("BASELINE-size cases run on the default policy only")
    if force_dma == 2:      # the 16 x 32-tile instantiations of the 256-pixel kernel forced: a dual launch keeps its 256-pixel tiles
        if H % 32 or W % 16:
            pytest.skip("32 x 16-pixel tiles only")
        policy(large_maps=0, big_tiles=2)
    elif force_dma:
        if H % 16 or W % 32:
            pytest.skip("16 x 32-pixel tiles only")
        policy(large_maps=2)
    K = K1 + K2
    xa, xb = RNG.standard_normal((B, H, W, K1)), RNG.standard_normal((B, H, W, K2))
    w = RNG.standard_normal((3, 3, K, N)) / np.sqrt(9 * K)
    dy = RNG.standard_normal((B, H, W, N))
    xad, xbd, wd, dyd = dev(xa, BF16), dev(xb, BF16), dev(w), dev(dy, BF16)
    xcd = torch.cat([xad, xbd], dim=3).contiguous()
    wf = torch.empty(9 * N * K, dtype=torch.bfloat16).cuda()
    wg = torch.empty(9 * N * K, dtype=torch.bfloat16).cuda()
    L.pack_conv3x3_bf16(wd.data_ptr(), wf.data_ptr(), wg.data_ptr(), K, N, S())

    def ws_for(k, n):
        nb = int(L.conv3x3_mfma_ws_bytes(B, H, W, k, n))
        t = torch.empty(max(nb // 4, 1), dtype=torch.float32).cuda()
        return (t.data_ptr() if nb else None), nb, t
    # forward
    wp, wb, _k1 = ws_for(K, N)
    y_ref, y = torch.empty(B, H, W, N, dtype=torch.bfloat16).cuda(), torch.empty(B, H, W, N, dtype=torch.bfloat16).cuda()
    L.conv3x3_mfma_bf16_ws(xcd.data_ptr(), wf.data_ptr(), y_ref.data_ptr(), None, 0, None, wp, wb, B, H, W, K, N, S())
    L.conv3x3_mfma_bf16_dual(xad.data_ptr(), xbd.data_ptr(), K1, wf.data_ptr(), y.data_ptr(), None, 0, None, None, 0, None, 0, wp, wb,
                             B, H, W, K, N, S())
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref)
    if B * H * W <= 65536:
        ref = O.conv2d_same(torch.cat([rounded(xa, BF16), rounded(xb, BF16)], dim=3), rounded(w, BF16))
        close(host(y), ref.numpy(), 1.5e-2, "dual forward vs oracle concat + conv")
    # ... with bias + ReLU, and with the statistics epilogue of the batch-norm layers (partial rows; atomics where few tiles exist)
    bias = dev(RNG.standard_normal(N) * 0.3)
    yb_ref, yb = torch.empty_like(y), torch.empty_like(y)
    L.conv3x3_mfma_bf16_ws(xcd.data_ptr(), wf.data_ptr(), yb_ref.data_ptr(), bias.data_ptr(), 1, None, wp, wb, B, H, W, K, N, S())
    L.conv3x3_mfma_bf16_dual(xad.data_ptr(), xbd.data_ptr(), K1, wf.data_ptr(), yb.data_ptr(), None, 0, bias.data_ptr(), None, 1, None, 0,
                             wp, wb, B, H, W, K, N, S())
    torch.cuda.synchronize()
    assert torch.equal(yb, yb_ref)
    ntile, ntile_d = L.conv3x3_mfma_bf16_tiles(B, H, W, K, N), L.conv3x3_mfma_bf16_tiles_dual(B, H, W, K, N)
    part_ref, part = torch.zeros(ntile, 2, N).cuda(), torch.zeros(ntile_d + 1, 2, N).cuda()
    L.conv3x3_mfma_bf16(xcd.data_ptr(), wf.data_ptr(), yb_ref.data_ptr(), None, 0, part_ref.data_ptr(), B, H, W, K, N, S())
    L.conv3x3_mfma_bf16_dual(xad.data_ptr(), xbd.data_ptr(), K1, wf.data_ptr(), yb.data_ptr(), None, 0, None, None, 0, part.data_ptr(), 1,
                             None, 0, B, H, W, K, N, S())
    torch.cuda.synchronize()
    assert torch.equal(yb, yb_ref)
    close(host(part)[:ntile_d].sum(0), host(part_ref).sum(0), 2e-5, "dual input, partial-row statistics")
    assert float(part[ntile_d].abs().max()) == 0.0            # nothing behind the rows phx_conv3x3_mfma_bf16_tiles_dual counts
    if L.conv3x3_mfma_stats_atomic_supported(B, H, W, K, N):
        sums = torch.zeros(N, 2).cuda()
        L.conv3x3_mfma_bf16_dual(xad.data_ptr(), xbd.data_ptr(), K1, wf.data_ptr(), yb.data_ptr(), None, 0, None, None, 0, sums.data_ptr(), 2,
                                 None, 0, B, H, W, K, N, S())
        close(host(sums).T, host(part_ref).sum(0), 2e-5, "dual input, atomic statistics")
    # data gradient: d(concat) = conv(dy, flipped filter), written as two tensors
    wp, wb, _k2 = ws_for(N, K)
    dx_ref = torch.empty(B, H, W, K, dtype=torch.bfloat16).cuda()
    L.conv3x3_mfma_bf16_ws(dyd.data_ptr(), wg.data_ptr(), dx_ref.data_ptr(), None, 0, None, wp, wb, B, H, W, N, K, S())
    g1 = torch.full((B, H, W, K1), 7.0, dtype=torch.bfloat16).cuda()
    g2 = torch.full((B, H, W, K2), 7.0, dtype=torch.bfloat16).cuda()
    L.conv3x3_mfma_bf16_dual(dyd.data_ptr(), None, 0, wg.data_ptr(), g1.data_ptr(), g2.data_ptr(), K1, None, None, 0, None, 0, wp, wb,
                             B, H, W, N, K, S())
    torch.cuda.synchronize()
    assert torch.equal(g1, dx_ref[..., :K1]) and torch.equal(g2, dx_ref[..., K1:])
    # filter gradient
    nb = int(L.conv3x3_wgrad_ws_bytes(B, H, W, K, N))
    nbd = int(L.conv3x3_wgrad_ws_bytes_dual(B, H, W, K, N, K1))
    wsr, wsd = torch.empty(max(nb // 4, 1), dtype=torch.float32).cuda(), torch.empty(max(nbd // 4, 1), dtype=torch.float32).cuda()
    dw_ref, dw = torch.zeros(9 * K * N, dtype=torch.float32).cuda(), torch.zeros(9 * K * N, dtype=torch.float32).cuda()
    L.conv3x3_wgrad_mfma_bf16(xcd.data_ptr(), dyd.data_ptr(), dw_ref.data_ptr(), wsr.data_ptr(), nb, B, H, W, K, N, S())
    L.conv3x3_wgrad_mfma_bf16_dual(xad.data_ptr(), xbd.data_ptr(), K1, dyd.data_ptr(), dw.data_ptr(), wsd.data_ptr(), nbd, B, H, W, K, N, 1, S())
    close(host(dw), host(dw_ref), 2e-5, "dual filter gradient")


@pytest.mark.parametrize("case", [(64, 16, 32), (5, 48, 96), (3, 16, 64), (2, 32, 32)])
def test_conv3x3_c32_against_the_general_kernel(Ld, case, policy):
    """k_conv3x3_c32 (32 -> 32 channels on large maps: filter in registers, persistent tiles; forced here on small maps) against the
    256-pixel kernel through the same entry points -- plain output, bias + activation epilogue, per-tile partial statistics -- and
    against the oracle's conv2d."""
    L = Ld                                   # (the test build: this test sets the kernel policy)
    B, H, W = case
    K = N = 32
    x = RNG.standard_normal((B, H, W, K))
    w = RNG.standard_normal((3, 3, K, N)) / np.sqrt(9 * K)
    bias = RNG.standard_normal(N) * 0.3
    xd, wd, bd = dev(x, BF16), dev(w), dev(bias)
    wf = torch.empty(9 * N * K, dtype=torch.bfloat16).cuda()
    wg = torch.empty(9 * N * K, dtype=torch.bfloat16).cuda()
    L.pack_conv3x3_bf16(wd.data_ptr(), wf.data_ptr(), wg.data_ptr(), K, N, S())
    res = {}
    for mode in (0, 2):
        policy(large_maps=mode)
        ntile = L.conv3x3_mfma_bf16_tiles(B, H, W, K, N)
        y, yb = (torch.empty(B, H, W, N, dtype=torch.bfloat16).cuda() for _ in range(2))
        part = torch.zeros(ntile, 2, N, dtype=torch.float32).cuda()
        L.conv3x3_mfma_bf16(xd.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, part.data_ptr(), B, H, W, K, N, S())
        L.conv3x3_mfma_bf16(xd.data_ptr(), wf.data_ptr(), yb.data_ptr(), bd.data_ptr(), 1, None, B, H, W, K, N, S())
        torch.cuda.synchronize()
        res[mode] = (host(y), host(yb), host(part).sum(0))
    a, b = res[0], res[2]
    close(b[0], a[0], 2 ** -7, "c32 vs 256-pixel kernel")            # (one chunk of 32 channels: same products, summation order of the taps)
    close(b[1], a[1], 2 ** -7, "c32 vs 256-pixel kernel, bias + relu")
    close(b[2], b[0].reshape(-1, N).sum(0)[None].repeat(2, 0) * 0 + np.stack([b[0].reshape(-1, N).sum(0), (b[0].reshape(-1, N) ** 2).sum(0)]), 2e-5,
          "partial-row statistics of the stored output")
    ref = T.conv2d_same(rounded(x, BF16), rounded(w, BF16))
    close(b[0], ref.numpy(), 1.5e-2, "c32 forward vs oracle")
    close(b[1], T.relu(T.bias_add(ref, torch.as_tensor(bias))).numpy(), 1.5e-2, "c32 bias + relu vs oracle")


@pytest.mark.parametrize("case", [(64, 8, 8, 192, 192, 12), (64, 4, 4, 64, 192, 12), (64, 2, 2, 192, 192, 12), (5, 16, 16, 64, 64, 4),
                                  (7, 8, 8, 32, 32, 32), (3, 16, 16, 64, 96, 96), (70, 2, 2, 32, 64, 4), (9, 4, 8, 32, 32, 2)])
@pytest.mark.parametrize("act", [1, 0])
def test_conv3x3_fused_group_norm_one_launch(L, case, act):
    """phx_conv3x3_mfma_bf16_fgn -- conv2d + bias + group norm (16-channel groups) / instance norm + activation in one launch
    (tfwrapper/layers.py:123-135 + normalisation.py:3-36): y bit-equal to the convolution with its bias epilogue; a, mean, rstd, scale,
    shift against the oracle's group_norm / instance_norm of that (bf16) y."""
    B, H, W, K, N, G = case
    assert L.conv3x3_fgn_supported(B, H, W, K, N, G) in (32, 64)
    assert L.conv3x3_fgn_supported(B, 32, 32, K, N, G) == 0 and L.conv3x3_fgn_supported(B, H, W, K, N, 5) == 0
    x = RNG.standard_normal((B, H, W, K)) + 0.2
    w = RNG.standard_normal((3, 3, K, N)) / np.sqrt(9 * K)
    bias = 0.3 * RNG.standard_normal(N)
    gamma, beta = 1.0 + 0.2 * RNG.standard_normal(N), 0.3 * RNG.standard_normal(N)
    xd, wd, bd, g_, b_ = dev(x, BF16), dev(w), dev(bias), dev(gamma), dev(beta)
    wf = torch.empty(9 * N * K, dtype=torch.bfloat16).cuda()
    wg = torch.empty(9 * N * K, dtype=torch.bfloat16).cuda()
    L.pack_conv3x3_bf16(wd.data_ptr(), wf.data_ptr(), wg.data_ptr(), K, N, S())
    y0 = torch.empty(B, H, W, N, dtype=torch.bfloat16).cuda()
    L.conv3x3_mfma_bf16_ws(xd.data_ptr(), wf.data_ptr(), y0.data_ptr(), bd.data_ptr(), 0, None, None, 0, B, H, W, K, N, S())
    y, a = torch.empty_like(y0), torch.empty_like(y0)
    mean, rstd = torch.empty(B, G, dtype=torch.float32).cuda(), torch.empty(B, G, dtype=torch.float32).cuda()
    scale, shift = torch.empty(B, N, dtype=torch.float32).cuda(), torch.empty(B, N, dtype=torch.float32).cuda()
    eps = 1e-5
    L.conv3x3_mfma_bf16_fgn(xd.data_ptr(), wf.data_ptr(), y.data_ptr(), a.data_ptr(), bd.data_ptr(), g_.data_ptr(), b_.data_ptr(), eps, G, act,
                            mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), B, H, W, K, N, S())
    torch.cuda.synchronize()
    assert torch.equal(y, y0)
    yf = torch.as_tensor(host(y))
    if G == N:
        ref = T.instance_norm(yf, torch.as_tensor(gamma), torch.as_tensor(beta), eps)
    else:
        ref = T.group_norm(yf, torch.as_tensor(gamma), torch.as_tensor(beta), G, eps)
    if act == 1:
        ref = T.relu(ref)
    close(host(a), ref.numpy(), 1.2e-2, "a = act(gn(conv + bias))")               # bf16 output
    yg = yf.numpy().reshape(B, H * W, G, N // G)
    mu, var = yg.mean(axis=(1, 3)), yg.var(axis=(1, 3))
    close(host(mean), mu, 2e-5, "mean")
    close(host(rstd), 1.0 / np.sqrt(var + eps), 2e-4, "rstd")
    sc_ref = np.repeat(1.0 / np.sqrt(var + eps), N // G, axis=1) * gamma
    close(host(scale), sc_ref, 2e-4, "scale")
    close(host(shift), beta - np.repeat(mu, N // G, axis=1) * sc_ref, 5e-4, "shift")


@pytest.mark.parametrize("C", [32, 192])
def test_norm_reduce_partials_every_tail(L, C):
    """phx_norm_reduce_partials / _ns (per-tile statistics rows of the convolution epilogue -> sums[c][2] / sums[ns][c][2]): the
    four-chain loops and their remainders -- tile counts below, at and off the multiples of 4 x 256 (batch form: 256 threads stride
    the tiles) and of 4 (per-sample form: a thread walks its sample's tiles)."""
    for ntile in (1, 3, 255, 256, 257, 1024, 1030, 4099):
        part = RNG.standard_normal((ntile, 2, C))
        sums = torch.full((C, 2), 7.0, dtype=torch.float32).cuda()
        L.norm_reduce_partials(dev(part).data_ptr(), ntile, C, sums.data_ptr(), S())
        close(host(sums), np.float32(part).astype(np.float64).sum(axis=0).T, 2e-6 * np.sqrt(ntile) + 1e-6, "reduce_partials T=%d" % ntile)
    NS = 3
    for T_ in (1, 2, 4, 6, 7, 64, 67):
        part = RNG.standard_normal((NS, T_, 2, C))
        sums = torch.full((NS, C, 2), 7.0, dtype=torch.float32).cuda()
        L.norm_reduce_partials_ns(dev(part).data_ptr(), T_, NS, C, sums.data_ptr(), S())
        ref = np.float32(part).astype(np.float64).sum(axis=1).transpose(0, 2, 1)
        close(host(sums), ref, 2e-6 * np.sqrt(T_) + 1e-6, "reduce_partials_ns T=%d" % T_)


def test_conv3x3_split_k_finish_slice_counts(L):
    """k_splitk_finish sums its fp32 slices four at a time (loads first) with remainders of two and one: shapes whose split-K
    factor covers 2 .. 9+ slices, against the unsplit launch of the same convolution (bf16 outputs: one rounding apart at most,
    the fp32 sums differ by summation order only), with bias + ReLU and without."""
    seen = set()
    for (B, H, W, K, N) in [(1, 4, 4, 64, 32), (2, 4, 4, 96, 32), (2, 4, 4, 128, 64), (1, 8, 8, 160, 32), (2, 8, 8, 192, 192),
                            (1, 2, 2, 192, 64), (2, 2, 2, 288, 32), (1, 4, 4, 384, 64), (1, 8, 8, 576, 64), (4, 16, 16, 192, 192)]:
        ks = int(L.conv3x3_mfma_ksplit(B, H, W, K, N))
        seen.add(ks)
        x = RNG.standard_normal((B, H, W, K))
        w = RNG.standard_normal((3, 3, K, N)) / np.sqrt(9 * K)
        b = RNG.standard_normal(N) * 0.3
        xd, wd, bd = dev(x, BF16), dev(w), dev(b)
        wf = torch.empty(9 * N * K, dtype=torch.bfloat16).cuda()
        wg = torch.empty(9 * N * K, dtype=torch.bfloat16).cuda()
        L.pack_conv3x3_bf16(wd.data_ptr(), wf.data_ptr(), wg.data_ptr(), K, N, S())
        wsb = int(L.conv3x3_mfma_ws_bytes(B, H, W, K, N))
        wsk = torch.empty(max(wsb // 4, 1), dtype=torch.float32).cuda()
        for (bp, act) in ((None, 0), (bd.data_ptr(), 1)):
            y1 = torch.empty(B, H, W, N, dtype=torch.bfloat16).cuda()
            y2 = torch.empty(B, H, W, N, dtype=torch.bfloat16).cuda()
            L.conv3x3_mfma_bf16(xd.data_ptr(), wf.data_ptr(), y1.data_ptr(), bp, act, None, B, H, W, K, N, S())
            L.conv3x3_mfma_bf16_ws(xd.data_ptr(), wf.data_ptr(), y2.data_ptr(), bp, act, None, wsk.data_ptr(), wsb, B, H, W, K, N, S())
            close(host(y2), host(y1), 5e-3, "split-K (%d slices) vs unsplit, %s" % (ks, (B, H, W, K, N)))
    assert len([k for k in seen if k > 1]) >= 3, sorted(seen)       # (the shapes above are meant to reach several slice counts)


# the anti-phase pair kernel k_conv3x3_pp (conv_pp.hip; policy: large maps), forced on small shapes.  A persistent grid of 1 / 3 blocks makes a block
# walk several (tile pair, channel block) work items (persistent pipeline across tile boundaries, epilogue in the partner's matrix
# phase); odd tile counts leave the last pair's second half without a tile; N % 64 == 32 takes the 32-channel-block instantiation
@pytest.mark.parametrize("case", [(2, 16, 32, 32, 128), (1, 32, 64, 96, 64), (3, 16, 32, 32, 192), (1, 48, 32, 64, 256),
                                  (1, 16, 64, 160, 128), (2, 32, 32, 32, 32), (1, 16, 64, 192, 32), (3, 16, 32, 64, 96),
                                  (5, 16, 32, 64, 64), (1, 16, 32, 32, 64)])
@pytest.mark.parametrize("grid", [0, 1, 3])
def test_conv3x3_mfma_pair_kernel(Ld, case, grid, policy):
    L = Ld                                   # (the test build: this test sets the kernel policy)
    policy(large_maps=2, pair_grid=grid)
    _mfma_case(L, case)


@pytest.mark.parametrize("case", [(2, 16, 16, 32, 32, 0), (3, 16, 32, 64, 32, 1), (1, 32, 16, 96, 64, 0), (5, 8, 8, 32, 32, 1), (2, 4, 4, 32, 32, 0)])
def test_upsample_conv_phase_form_vs_oracle(L, case):
    """bilinear_upsample2D -> conv2D 3x3 SAME (tfwrapper/layers.py:336-345 into :123) WITHOUT the up-sampled tensor (csrc/upconv.hip +
    the ordinary matrix launches, phiseg_code_amd/upconv.py): the forward map, the gradient with respect to the LOW-resolution input
    and the filter gradient against the oracle's composition conv2d_same(resize_bilinear_legacy(x), W) and its autograd, on bf16 inputs.
    Tolerances: bf16 rounding of the stored output / input gradient and of the effective filters (two roundings: 1.2e-2 of the largest
    value), 2e-3 for the fp32 filter gradient of bf16 operands whose frame part passes through bf16 up-sampled rows."""
    from oracle import tf1_ops as T
    from phiseg_code_amd import upconv
    B, h, w, K, N, with_bias = case                     # (with_bias: the convolution bias group / instance norm layers keep, layers.py:126-132)
    rng = np.random.default_rng(5)
    bias = torch.as_tensor(rng.standard_normal(N) * 0.3, dtype=torch.float32)
    x = rounded(rng.standard_normal((B, h, w, K)), BF16)
    W = torch.as_tensor(rng.standard_normal((3, 3, K, N)) / np.sqrt(9 * K), dtype=torch.float32)
    dy = rounded(rng.standard_normal((B, 2 * h, 2 * w, N)) * 0.1, BF16)
    xr, wr = x.clone().requires_grad_(True), W.double().clone().requires_grad_(True)
    ref = T.conv2d_same(T.resize_bilinear_legacy(xr, 2 * h, 2 * w), wr) + (bias.double() if with_bias else 0.0)
    ref.backward(dy)
    keep = []

    def alloc(shape, dt):
        t = torch.empty(int(np.prod(shape)), dtype=tdt(dt), device="cuda")
        keep.append(t)
        return type("B", (), dict(ptr=t.data_ptr(), t=t, shape=tuple(shape)))()

    def alloc_zeroed(n):
        t = torch.zeros(int(n), dtype=torch.float32, device="cuda")
        keep.append(t)
        return type("B", (), dict(ptr=t.data_ptr(), t=t))()

    def emit(fn, *args, **kw):
        fn(*args)
    xd, wd32 = dev(x.numpy(), BF16), W.contiguous().cuda()
    xb = type("B", (), dict(ptr=xd.data_ptr()))()
    wf, wg = alloc((9 * K * N,), BF16), alloc((9 * K * N,), BF16)
    L.pack_conv3x3_bf16(wd32.data_ptr(), wf.ptr, wg.ptr, K, N, S())
    yp = alloc((B, h, w, 4 * N), BF16)
    bd = bias.cuda()
    ctx = upconv.forward(emit, alloc, L, S(), xb, wd32.data_ptr(), wf, yp, B, h, w, K, N, bias_ptr=bd.data_ptr() if with_bias else None)
    yhi = alloc((B, 2 * h, 2 * w, N), BF16)
    L.depth_to_space2(yp.ptr, yhi.ptr, B, h, w, N, S())
    close(host(yhi.t).reshape(B, 2 * h, 2 * w, N), ref.detach().numpy(), 1.2e-2, "forward")
    # backward: dy in the packed pixel order
    dyd = dev(dy.numpy(), BF16)
    dyp = alloc((B, h, w, 4 * N), BF16)
    L.space_to_depth2(dyd.data_ptr(), dyp.ptr, B, h, w, N, S())
    back = alloc((B, 2 * h, 2 * w, N), BF16)
    L.depth_to_space2(dyp.ptr, back.ptr, B, h, w, N, S())
    assert torch.equal(back.t.reshape(-1), dyd.reshape(-1))             # the two permutations are inverses
    dw = torch.full((3, 3, K, N), 0.25, dtype=torch.float32, device="cuda")      # accumulate semantics
    upconv.backward_prepare(emit, alloc, L, S(), ctx, dyp, B, h, w, N)
    upconv.backward_filters(emit, alloc, alloc_zeroed, L, S(), ctx, xb, dyp, dw.data_ptr(), B, h, w, K, N)
    dx = alloc((B, h, w, K), BF16)
    upconv.backward_data(emit, alloc, L, S(), ctx, dyp, wg, dx, B, h, w, K, N)
    close(host(dw) - 0.25, wr.grad.numpy(), 2e-3, "filter gradient")
    close(host(dx.t).reshape(B, h, w, K), xr.grad.numpy(), 1.2e-2, "gradient with respect to the low-resolution input")


@pytest.mark.parametrize("case", [(3, 8, 8, 32, "batch"), (2, 16, 4, 64, "batch"), (5, 4, 12, 192, "batch"), (3, 8, 4, 64, "group"),
                                  (2, 4, 8, 32, "instance")])
def test_norm_passes_with_the_pixel_permutation_of_the_phase_form(L, case):
    """The normalisation layer behind a phase-form convolution (csrc/upconv.hip): y is stored in the packed pixel order [B, h, w, (a, b)],
    its readers want hi-res rows.  phx_norm_apply_fused_d2s == phx_norm_apply_fused followed by phx_depth_to_space2 (bit for bit);
    phx_norm_bwd_reduce_s2d / phx_norm_bwd_apply_fused_s2d == phx_space_to_depth2 of dA followed by the plain passes (sums to 1e-5:
    the reduction's atomics; dy and the closed-form bias gradient to the last bit of that).  Batch norm (one statistic over all images)
    and group / instance norm (one per image, bias kept)."""
    B, h, w, C, norm = case
    NS, P, G = (1, B * 4 * h * w, C) if norm == "batch" else (B, 4 * h * w, C if norm == "instance" else C // 8)
    NP = NS * P
    g = torch.Generator(device="cuda").manual_seed(3)
    y = torch.randn(NP, C, device="cuda", generator=g).to(torch.bfloat16)
    gamma = (1.0 + 0.2 * torch.randn(C, device="cuda", generator=g)).float()
    beta = (0.1 * torch.randn(C, device="cuda", generator=g)).float()
    dA_hi = torch.randn(B, 2 * h, 2 * w, C, device="cuda", generator=g).to(torch.bfloat16)
    sums, piv = torch.zeros(NS * C * 2, device="cuda"), torch.zeros(NS * C, device="cuda")
    L.norm_stats(y.data_ptr(), BF16, sums.data_ptr(), piv.data_ptr(), NS, P, C, S())
    f = lambda: [torch.zeros(NS * G, device="cuda"), torch.zeros(NS * G, device="cuda"), torch.zeros(NS * C, device="cuda"), torch.zeros(NS * C, device="cuda")]
    m1, r1, sc1, sh1 = f()
    m2, r2, sc2, sh2 = f()
    a_p, a_hi, a_ref = [torch.empty(NP, C, device="cuda", dtype=torch.bfloat16) for _ in range(3)]
    L.norm_apply_fused(y.data_ptr(), BF16, sums.data_ptr(), piv.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-3, a_p.data_ptr(), BF16,
                       m1.data_ptr(), r1.data_ptr(), sc1.data_ptr(), sh1.data_ptr(), None, None, 0.0, NS, P, C, G, 1, S())
    L.depth_to_space2(a_p.data_ptr(), a_ref.data_ptr(), B, h, w, C, S())
    L.norm_apply_fused_d2s(y.data_ptr(), BF16, sums.data_ptr(), piv.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-3, a_hi.data_ptr(), BF16,
                           m2.data_ptr(), r2.data_ptr(), sc2.data_ptr(), sh2.data_ptr(), None, None, 0.0, NS, P, C, G, 1, h, w, S())
    torch.cuda.synchronize()
    assert torch.equal(a_hi, a_ref) and torch.equal(sc1, sc2) and torch.equal(sh1, sh2)
    # backward
    dA_p = torch.empty(NP, C, device="cuda", dtype=torch.bfloat16)
    L.space_to_depth2(dA_hi.data_ptr(), dA_p.data_ptr(), B, h, w, C, S())
    s_ref, s_got = torch.zeros(NS * C * 2, device="cuda"), torch.zeros(NS * C * 2, device="cuda")
    L.norm_bwd_reduce(dA_p.data_ptr(), BF16, y.data_ptr(), BF16, sc1.data_ptr(), sh1.data_ptr(), m1.data_ptr(), r1.data_ptr(), s_ref.data_ptr(),
                      NS, P, C, G, 1, 1, S())
    L.norm_bwd_reduce_s2d(dA_hi.data_ptr(), BF16, y.data_ptr(), BF16, sc1.data_ptr(), sh1.data_ptr(), m1.data_ptr(), r1.data_ptr(),
                          s_got.data_ptr(), NS, P, C, G, 1, 1, h, w, S())
    close(host(s_got), host(s_ref), 1e-5, "backward sums")
    dy_ref, dy_got = torch.empty_like(dA_p), torch.empty_like(dA_p)
    dg1, db1, dg2, db2, dbias1, dbias2 = [torch.zeros(C, device="cuda") for _ in range(6)]
    withb = norm != "batch"
    L.norm_bwd_apply_fused_bias(dA_p.data_ptr(), BF16, y.data_ptr(), BF16, sc1.data_ptr(), sh1.data_ptr(), m1.data_ptr(), r1.data_ptr(),
                                gamma.data_ptr(), s_ref.data_ptr(), dy_ref.data_ptr(), BF16, dg1.data_ptr(), db1.data_ptr(),
                                sums.data_ptr() if withb else None, piv.data_ptr() if withb else None, dbias1.data_ptr() if withb else None,
                                NS, P, C, G, 1, 1, S())
    L.norm_bwd_apply_fused_s2d(dA_hi.data_ptr(), BF16, y.data_ptr(), BF16, sc1.data_ptr(), sh1.data_ptr(), m1.data_ptr(), r1.data_ptr(),
                               gamma.data_ptr(), s_ref.data_ptr(), dy_got.data_ptr(), BF16, dg2.data_ptr(), db2.data_ptr(),
                               sums.data_ptr() if withb else None, piv.data_ptr() if withb else None, dbias2.data_ptr() if withb else None,
                               NS, P, C, G, 1, 1, h, w, S())
    torch.cuda.synchronize()
    assert torch.equal(dy_got, dy_ref)
    close(host(dg2), host(dg1), 1e-6, "dgamma")          # (atomics over the sample groups)
    close(host(db2), host(db1), 1e-6, "dbeta")
    close(host(dbias2), host(dbias1), 1e-6, "dbias")
