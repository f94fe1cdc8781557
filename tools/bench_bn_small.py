"""Micro-benchmark (GPU box): one-launch batch-norm layers (phx_bn_small_*) on the H <= 4 levels at batch 64, back to back."""
import sys
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
t = torch.zeros(256, device="cuda")
print("floor: add_ of 256 floats %.2f us" % timeit(lambda: t.add_(1.0)))
for (P, C) in [(256, 192), (1024, 192), (1024, 384), (4096, 192)]:
    x = torch.randn(P, C, device="cuda").to(torch.bfloat16)
    dA = torch.randn(P, C, device="cuda").to(torch.bfloat16)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    mean, rstd, scale, shift = (torch.empty(C, device="cuda") for _ in range(4))
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    f = lambda: L.bn_small_fwd(x.data_ptr(), 1, gamma.data_ptr(), beta.data_ptr(), 1e-3, y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                               scale.data_ptr(), shift.data_ptr(), None, None, 0.0, P, C, 1, st)
    b = lambda: L.bn_small_bwd(dA.data_ptr(), x.data_ptr(), 1, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                               gamma.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), P, C, 1, st)
    f()
    print("P=%5d C=%3d | fwd %6.2f us | bwd %6.2f us" % (P, C, timeit(f), timeit(b)), flush=True)
