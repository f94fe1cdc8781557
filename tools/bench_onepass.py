#!/usr/bin/env python3
"""Each launch alone: batch-norm backward in one launch (phx_bn_bwd_onepass) against phx_norm_bwd_reduce + phx_norm_bwd_apply_fused."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phiseg_code_amd import runtime as rt  # noqa: E402

L = rt.lib()
S = torch.cuda.current_stream().cuda_stream
nbar = int(L.bn_bwd_onepass_barrier_words())
for B, H, C in [(64, 8, 192), (64, 8, 128), (64, 16, 192), (64, 16, 64), (64, 16, 384), (64, 32, 32), (64, 32, 64), (64, 32, 128), (64, 32, 192),
                (64, 64, 64)]:
    P = B * H * H
    x = torch.randn(P, C, device="cuda").to(torch.bfloat16)
    dA = torch.randn(P, C, device="cuda").to(torch.bfloat16)
    dx = torch.empty_like(x)
    v = [torch.rand(C, device="cuda") + 0.5 for _ in range(5)]
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    sums = torch.zeros(4 * C * 2, device="cuda")
    bars = torch.zeros(64, nbar, dtype=torch.int32, device="cuda")

    def two():
        L.norm_bwd_reduce(dA.data_ptr(), 1, x.data_ptr(), 1, v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), v[3].data_ptr(), sums.data_ptr(), 1, P, C, C, 1, 4, S)
        L.norm_bwd_apply_fused(dA.data_ptr(), 1, x.data_ptr(), 1, v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), v[3].data_ptr(), v[4].data_ptr(), sums.data_ptr(),
                               dx.data_ptr(), 1, dg.data_ptr(), db.data_ptr(), 1, P, C, C, 1, 4, S)

    def t(fn, n):
        fn(0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn(i + 1)
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / n
    t2 = t(lambda i: two(), 20)
    if L.bn_bwd_onepass_supported(P, C, 1):
        bars.zero_()
        t1 = t(lambda i: L.bn_bwd_onepass(dA.data_ptr(), x.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), v[3].data_ptr(), v[4].data_ptr(), sums.data_ptr(),
                                          bars[i].data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), P, C, 1, 4, S), 20)
        assert int(bars[:, 288].sum()) == 0
    else:
        t1 = float("nan")
    print("%2d x %3d x %3d x %3d  (%5.1f MB per tensor)   two launches %6.1f us   one launch %6.1f us" % (B, H, H, C, P * C * 2 / 1e6, t2, t1), flush=True)
