#!/usr/bin/env python3
"""Each launch alone: the fp32 matrix-instruction convolution (csrc/conv_f32_mfma.hip) -- forward, data gradient, filter gradient --
against the direct fp32 kernels (csrc/conv_direct.hip) on the shape classes of phiseg_7_5 at batch 64.  TFLOP/s and fraction of the
157.3 TFLOP/s fp32 matrix peak.   python tools/bench_f32_mfma.py [--direct]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phiseg_code_amd import runtime as rt  # noqa: E402

L = rt.lib()
S = torch.cuda.current_stream().cuda_stream
SHAPES = [(64, 128, 128, 128, 128), (64, 128, 128, 64, 128), (64, 128, 128, 192, 32), (64, 128, 128, 32, 32), (64, 64, 64, 192, 192),
          (64, 64, 64, 64, 64), (64, 32, 32, 128, 128), (64, 32, 32, 256, 192), (64, 16, 16, 192, 192), (64, 8, 8, 192, 192),
          (64, 4, 4, 192, 192), (64, 2, 2, 192, 192), (64, 128, 128, 3, 32)]
direct = "--direct" in sys.argv


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
for B, H, W, Cin, Cout in SHAPES:
    x = torch.randn(B, H, W, Cin, device="cuda")
    dy = torch.randn(B, H, W, Cout, device="cuda")
    w = torch.randn(3, 3, Cin, Cout, device="cuda") / (3 * Cin ** 0.5)
    y = torch.empty(B, H, W, Cout, device="cuda")
    dx = torch.empty(B, H, W, Cin, device="cuda")
    dw = torch.zeros(3, 3, Cin, Cout, device="cuda")
    wf = torch.empty(int(L.conv3x3_f32_mfma_packed_floats(Cin, Cout)), device="cuda")
    wd = torch.empty(int(L.conv3x3_f32_mfma_packed_floats(Cout, Cin)), device="cuda")
    rec = np.zeros(1, dtype=[("w", "<u8"), ("wf", "<u8"), ("wd", "<u8"), ("cin", "<i4"), ("cout", "<i4")])
    rec[0] = (w.data_ptr(), wf.data_ptr(), wd.data_ptr() if Cin % 32 == 0 else 0, Cin, Cout)
    desc = torch.from_numpy(rec.view(np.uint8).copy()).cuda()
    L.pack_conv3x3_f32_multi(desc.data_ptr(), 1, S)
    wsb = int(L.conv3x3_f32_mfma_wgrad_ws_bytes(B, H, W, Cin, Cout, 0))
    ws = torch.empty(max(wsb // 4, 1), device="cuda")
    fl = 18.0 * Cin * Cout * B * H * W
    row = "%3d x %3d x %3d  %3d -> %3d " % (B, H, W, Cin, Cout)
    jobs = [("fwd", lambda: L.conv3x3_f32_mfma(x.data_ptr(), wf.data_ptr(), None, y.data_ptr(), B, H, W, Cin, Cout, 0, S))]
    if Cin % 32 == 0:
        jobs.append(("dgrad", lambda: L.conv3x3_f32_mfma(dy.data_ptr(), wd.data_ptr(), None, dx.data_ptr(), B, H, W, Cout, Cin, 0, S)))
    jobs.append(("wgrad", lambda: L.conv3x3_f32_mfma_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), wsb, B, H, W, Cin, Cout, S)))
    for name, fn in jobs:
        ms = timed(fn)
        tot[name][0] += fl
        tot[name][1] += ms
        row += " | %s %8.3f ms %6.1f TF %4.2f" % (name, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3)
    if direct:
        ms = timed(lambda: L.conv2d_direct(x.data_ptr(), 0, w.data_ptr(), None, y.data_ptr(), 0, B, H, W, Cin, Cout, 3, 0, 0, None, S), 2)
        row += " | direct fwd %8.3f ms %5.1f TF" % (ms, fl / ms / 1e9)
        ms = timed(lambda: L.conv2d_direct_wgrad(x.data_ptr(), 0, dy.data_ptr(), 0, dw.data_ptr(), None, B, H, W, Cin, Cout, 3, S), 2)
        row += " wgrad %8.3f ms %5.1f TF" % (ms, fl / ms / 1e9)
    print(row, flush=True)
for k, (fl, ms) in tot.items():
    print("%s: %.1f TFLOP/s over the table (%.2f of 157.3)" % (k, fl / ms / 1e9, fl / ms / 1e9 / 157.3))
