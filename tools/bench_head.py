"""Micro-benchmark (GPU box): 1x1 head filter-gradient kernel over the PHiSeg head shapes (B = 64)."""
import sys, os
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
for (H, C) in [(128, 32), (64, 64), (32, 128), (16, 192), (8, 192), (4, 192)]:
    npix = 64 * H * H
    x = torch.randn(npix, C, device="cuda").to(torch.bfloat16)
    dy = torch.randn(npix, 2, device="cuda")
    dw = torch.zeros(C * 2, device="cuda"); db = torch.zeros(2, device="cuda")
    fn = lambda: L.head1x1_wgrad(x.data_ptr(), rt.BF16, dy.data_ptr(), dw.data_ptr(), db.data_ptr(), npix, C, 2, st)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print("H=%-4d C=%-4d %6.1f us %6.0f GB/s" % (H, C, us, npix * (C * 2 + 8) / us / 1e3))
