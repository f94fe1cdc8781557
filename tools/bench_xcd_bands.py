"""GPU box: the kernels whose items share halos / rows with their neighbours, each launch alone (HIP events, batch 64), for the
library PHX_LIB names -- run once per build (tools/build_variant.sh rr -DPHX_TILE_BANDS=0 against the product library) to see what
the XCD-banded tile order (phx_band8, xcd_banded_block) does to each of them stand-alone."""
import sys
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
BF = rt.BF16


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def conv(B, H, W, K, N):
    x = torch.randn(B, H, W, K, device="cuda").to(torch.bfloat16)
    wf = (torch.randn(9 * K * N, device="cuda") * 0.05).to(torch.bfloat16)
    y = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16)
    nt = L.conv3x3_mfma_bf16_tiles(B, H, W, K, N)
    part = torch.zeros(nt * 2 * N, device="cuda")
    f = lambda: L.conv3x3_mfma_bf16(x.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, part.data_ptr(), B, H, W, K, N, st)
    return timeit(f), (x.numel() + y.numel()) * 2


def resize(B, h, w, C):
    x = torch.randn(B, h, w, C, device="cuda").to(torch.bfloat16)
    y = torch.empty(B, 2 * h, 2 * w, C, device="cuda", dtype=torch.bfloat16)
    f = lambda: L.bilinear_up2x_fwd(x.data_ptr(), BF, y.data_ptr(), B, h, w, C, st)
    g = lambda: L.bilinear_up2x_bwd(y.data_ptr(), BF, x.data_ptr(), B, h, w, C, st)
    return timeit(f), timeit(g), (x.numel() + y.numel()) * 2


print("library:", rt.LIB_PATH)
for shp in [(64, 128, 128, 32, 32), (64, 128, 128, 128, 128), (64, 128, 128, 192, 32), (64, 64, 64, 192, 192), (64, 32, 32, 192, 192), (64, 32, 32, 128, 128)]:
    t, nb = conv(*shp)
    print("conv3x3 %3d -> %3d @ %3d x %3d: %7.1f us  (%.2f TB/s of tensor bytes, %.0f TFLOP/s)" % (shp[3], shp[4], shp[1], shp[2], t, nb / t / 1e6, 2 * 9 * shp[0] * shp[1] * shp[2] * shp[3] * shp[4] / t / 1e6))
for shp in [(64, 64, 64, 192), (64, 32, 32, 192), (64, 64, 64, 32)]:
    tf, tb, nb = resize(*shp)
    print("resize x2 %3d ch from %3d x %3d: forward %6.1f us (%.2f TB/s)  adjoint %6.1f us (%.2f TB/s)" % (shp[3], shp[1], shp[2], tf, nb / tf / 1e6, tb, nb / tb / 1e6))
