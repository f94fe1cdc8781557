"""dev (GPU box): the norm + 1x1 head fusion at the likelihood's top layer (64 x 128 x 128 x 128 -> 2): separate launches vs fused."""
import sys
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
BF = rt.BF16
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, H, W, C, NO = 64, 128, 128, 128, 2
P = B * H * W
y = (torch.randn(P, C, device="cuda") * 1.5).to(torch.bfloat16)
a = torch.empty_like(y); dA = torch.empty_like(y); dx = torch.empty_like(y)
sums = torch.stack([y.float().sum(0), (y.float() ** 2).sum(0)], -1).contiguous()
g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
wh, bh = torch.randn(C, NO, device="cuda") * 0.1, torch.zeros(NO, device="cuda")
mean, rstd, scale, shift = (torch.empty(C, device="cuda") for _ in range(4))
yh, dyh = torch.empty(P, NO, device="cuda"), torch.randn(P, NO, device="cuda")
s2 = torch.zeros(4, C, 2, device="cuda"); dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
ap = lambda: L.norm_apply_fused(y.data_ptr(), BF, sums.data_ptr(), None, g.data_ptr(), b.data_ptr(), 1e-3, a.data_ptr(), BF, mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), None, None, 0.0, 1, P, C, C, 1, st)
hf = lambda: L.head1x1_fwd(a.data_ptr(), BF, wh.data_ptr(), bh.data_ptr(), yh.data_ptr(), P, C, NO, 0, st)
aph = lambda: L.norm_apply_fused_head(y.data_ptr(), BF, sums.data_ptr(), None, g.data_ptr(), b.data_ptr(), 1e-3, a.data_ptr(), BF, mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), None, None, 0.0, 1, P, C, C, 1, wh.data_ptr(), bh.data_ptr(), NO, yh.data_ptr(), st)
hd = lambda: L.head1x1_dgrad(dyh.data_ptr(), wh.data_ptr(), dA.data_ptr(), BF, P, C, NO, st)
br = lambda: L.norm_bwd_reduce(dA.data_ptr(), BF, y.data_ptr(), BF, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), s2.data_ptr(), 1, P, C, C, 1, 4, st)
brh = lambda: L.norm_bwd_reduce_head(dyh.data_ptr(), wh.data_ptr(), NO, y.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), s2.data_ptr(), 1, P, C, C, 1, 4, st)
ba = lambda: L.norm_bwd_apply_fused_bias(dA.data_ptr(), BF, y.data_ptr(), BF, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), s2.data_ptr(), dx.data_ptr(), BF, dg.data_ptr(), db.data_ptr(), None, None, None, 1, P, C, C, 1, 4, st)
bah = lambda: L.norm_bwd_apply_fused_head(dyh.data_ptr(), wh.data_ptr(), NO, y.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), s2.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), None, None, None, 1, P, C, C, 1, 4, st)
for name, fn in (("apply", ap), ("head fwd", hf), ("apply+head", aph), ("head dgrad", hd), ("bwd reduce", br), ("bwd reduce (head)", brh), ("bwd apply", ba), ("bwd apply (head)", bah)):
    print("%-20s %7.1f us" % (name, timeit(fn)), flush=True)
