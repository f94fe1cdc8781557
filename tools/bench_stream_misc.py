"""Micro-benchmark (GPU box): the small streaming kernels (1x1 heads forward / data gradient, in-place add, pooling) at the
model's shapes, batch 64, against a plain device copy of the same number of bytes."""
import sys
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
BF, F32 = rt.BF16, rt.F32
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (H, C) in [(128, 32), (64, 64), (32, 128), (16, 192), (8, 192)]:
    P = 64 * H * H
    x = torch.randn(P, C, device="cuda").to(torch.bfloat16)
    g = torch.randn(P, C, device="cuda").to(torch.bfloat16)
    w, b = torch.randn(C, 2, device="cuda"), torch.zeros(2, device="cuda")
    y, dy = torch.empty(P, 2, device="cuda"), torch.randn(P, 2, device="cuda")
    dx = torch.empty_like(x)
    half = torch.empty(P // 4, C, device="cuda", dtype=torch.bfloat16)
    mb = P * C * 2 / 1e6
    t_f = timeit(lambda: L.head1x1_fwd(x.data_ptr(), BF, w.data_ptr(), b.data_ptr(), y.data_ptr(), P, C, 2, 0, st))
    t_d = timeit(lambda: L.head1x1_dgrad(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), BF, P, C, 2, st))
    t_a = timeit(lambda: L.add_inplace(x.data_ptr(), g.data_ptr(), P * C, BF, st))
    t_p = timeit(lambda: L.avgpool2x2_fwd(x.data_ptr(), BF, half.data_ptr(), 64, H, H, C, st))
    t_q = timeit(lambda: L.avgpool2x2_bwd(half.data_ptr(), BF, dx.data_ptr(), 64, H, H, C, st))
    t_c = timeit(lambda: dx.copy_(x))
    print("H=%3d C=%3d %6.1f MB | head fwd %6.1f us %5.0f GB/s | head dgrad %6.1f us %5.0f GB/s | add %6.1f us %5.0f GB/s | pool fwd %6.1f bwd %6.1f us | copy %6.1f us"
          % (H, C, mb, t_f, mb / t_f * 1e3, t_d, mb / t_d * 1e3, t_a, 3 * mb / t_a * 1e3, t_p, t_q, t_c), flush=True)
# bilinear x2 up-sampling (TF1 legacy) and concat / split at the likelihood's top-down shapes
for (h, C) in [(64, 192), (32, 192), (16, 192), (8, 192)]:
    x = torch.randn(64, h, h, C, device="cuda").to(torch.bfloat16)
    y = torch.empty(64, 2 * h, 2 * h, C, device="cuda", dtype=torch.bfloat16)
    mb = (x.numel() + y.numel()) * 2 / 1e6
    t_u = timeit(lambda: L.bilinear_up2x_fwd(x.data_ptr(), BF, y.data_ptr(), 64, h, h, C, st))
    t_b = timeit(lambda: L.bilinear_up2x_bwd(y.data_ptr(), BF, x.data_ptr(), 64, h, h, C, st))
    print("bilinear %3d -> %3d C=%3d %6.1f MB | fwd %6.1f us %5.0f GB/s | bwd %6.1f us %5.0f GB/s" % (h, 2 * h, C, mb, t_u, mb / t_u * 1e3, t_b, mb / t_b * 1e3), flush=True)
for (H, Ca, Cb) in [(128, 32, 32), (64, 64, 64), (32, 128, 128), (16, 192, 192)]:
    P = 64 * H * H
    a = torch.randn(P, Ca, device="cuda").to(torch.bfloat16); b = torch.randn(P, Cb, device="cuda").to(torch.bfloat16)
    o = torch.empty(P, Ca + Cb, device="cuda", dtype=torch.bfloat16)
    mb = 2 * o.numel() * 2 / 1e6
    t_c = timeit(lambda: L.concat2(a.data_ptr(), Ca, b.data_ptr(), Cb, o.data_ptr(), P, BF, st))
    t_s = timeit(lambda: L.split2(o.data_ptr(), a.data_ptr(), Ca, b.data_ptr(), Cb, P, BF, st))
    print("concat H=%3d %d+%d %6.1f MB | concat %6.1f us %5.0f GB/s | split %6.1f us %5.0f GB/s" % (H, Ca, Cb, mb, t_c, mb / t_c * 1e3, t_s, mb / t_s * 1e3), flush=True)
