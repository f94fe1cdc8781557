"""Micro-benchmark (GPU box): the three bf16 batch-norm streaming kernels over the PHiSeg activation shapes (B = 64)."""
import sys
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
BF = rt.BF16
shapes = [(128, 32), (128, 128), (64, 64), (64, 192), (32, 128), (32, 192), (16, 192), (8, 192), (4, 192), (2, 192)]
for (H, C) in shapes:
    P = 64 * H * H
    x = torch.randn(P, C, device="cuda").to(torch.bfloat16)
    dA = torch.randn(P, C, device="cuda").to(torch.bfloat16)
    y = torch.empty_like(x); dx = torch.empty_like(x)
    f = lambda n: torch.zeros(n, device="cuda")
    sums = torch.stack([x.float().sum(0), (x.float() ** 2).sum(0)], 1).contiguous()
    gamma, beta = torch.ones(C, device="cuda"), f(C)
    mean, rstd, scale, shift, mm, mv = f(C), f(C), f(C), f(C), f(C), torch.ones(C, device="cuda")
    import os
    NREP = int(os.environ.get("BENCH_NREP", "8"))
    sums2, dg, db = f(NREP * 2 * C), f(C), f(C)
    def k_apply():
        L.norm_apply_fused(x.data_ptr(), BF, sums.data_ptr(), None, gamma.data_ptr(), beta.data_ptr(), 1e-3, y.data_ptr(), BF,
                           mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), mm.data_ptr(), mv.data_ptr(), 0.0,
                           1, P, C, C, 1, st)
    def k_bred():
        L.norm_bwd_reduce(dA.data_ptr(), BF, x.data_ptr(), BF, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                          sums2.data_ptr(), 1, P, C, C, 1, NREP, st)
    def k_bapp():
        L.norm_bwd_apply_fused(dA.data_ptr(), BF, x.data_ptr(), BF, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                               gamma.data_ptr(), sums2.data_ptr(), dx.data_ptr(), BF, dg.data_ptr(), db.data_ptr(), 1, P, C, C, 1, NREP, st)
    out = []
    for name, fn, bpe in (("apply", k_apply, 4), ("bwd_reduce", k_bred, 4), ("bwd_apply", k_bapp, 6)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        out.append("%s %6.1f us %5.0f GB/s" % (name, us, P * C * bpe / us / 1e3))
    print("H=%-4d C=%-4d %6.1f MB | " % (H, C, P * C * 2 / 1e6) + " | ".join(out))
    # yardstick: a plain device copy of the same tensor (read + write, 4 bytes per element)
    for _ in range(3): y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print("                         | copy_ %6.1f us %5.0f GB/s" % (us, P * C * 4 / us / 1e3))
