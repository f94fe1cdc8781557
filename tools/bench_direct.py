"""Micro-benchmark (GPU box): fp32-math direct conv on the narrow-head shapes of the bf16 plan."""
import sys
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
F32, BF = rt.F32, rt.BF16
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 50
for (B, H, Cin, Cout, k) in [(64, 2, 192, 2, 3), (64, 2, 192, 2, 1), (64, 128, 32, 2, 1)]:
    x = torch.randn(B, H, H, Cin, device="cuda").to(torch.bfloat16)
    w = torch.randn(k, k, Cin, Cout, device="cuda")
    b = torch.randn(Cout, device="cuda")
    y = torch.empty(B, H, H, Cout, device="cuda")
    dy = torch.randn(B, H, H, Cout, device="cuda")
    dx = torch.empty(B, H, H, Cin, device="cuda", dtype=torch.bfloat16)
    dw = torch.zeros(k, k, Cin, Cout, device="cuda"); db = torch.zeros(Cout, device="cuda")
    t1 = timeit(lambda: L.conv2d_direct(x.data_ptr(), BF, w.data_ptr(), b.data_ptr(), y.data_ptr(), F32, B, H, H, Cin, Cout, k, 0, 0, None, st))
    t2 = timeit(lambda: L.conv2d_direct(dy.data_ptr(), F32, w.data_ptr(), None, dx.data_ptr(), BF, B, H, H, Cin, Cout, k, 0, 1, None, st))
    t3 = timeit(lambda: L.conv2d_direct_wgrad(x.data_ptr(), BF, dy.data_ptr(), F32, dw.data_ptr(), db.data_ptr(), B, H, H, Cin, Cout, k, st))
    print((B, H, Cin, Cout, k), "fwd %.1f us  dgrad %.1f us  wgrad %.1f us" % (t1, t2, t3))
