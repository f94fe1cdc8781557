"""Micro-benchmark (GPU box): group-norm layer, one-launch kernels (phx_norm_small_*) vs the streaming chain
(norm_stats + norm_apply_fused; norm_bwd_reduce + norm_bwd_apply_fused_bias), batch 64, the model's map sizes."""
import sys
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
BF = rt.BF16 if hasattr(rt, "BF16") else 1
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 64
for (H, C) in [(2, 192), (4, 192), (8, 192), (16, 192), (16, 384)]:
    P, G = H * H, C // 16
    x = torch.randn(B, H, H, C, device="cuda").to(torch.bfloat16)
    dA = torch.randn(B, H, H, C, device="cuda").to(torch.bfloat16)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    mean, rstd = torch.empty(B * G, device="cuda"), torch.empty(B * G, device="cuda")
    scale, shift = torch.empty(B * C, device="cuda"), torch.empty(B * C, device="cuda")
    sums, pivot = torch.zeros(B, C, 2, device="cuda"), torch.zeros(B, C, device="cuda")
    nrep = 8 if P >= 4096 else 1
    sums2 = torch.zeros(nrep, B, C, 2, device="cuda")
    dg, db, dbias = (torch.zeros(C, device="cuda") for _ in range(3))
    def f_chain():
        L.norm_stats(x.data_ptr(), BF, sums.data_ptr(), pivot.data_ptr(), B, P, C, st)
        L.norm_apply_fused(x.data_ptr(), BF, sums.data_ptr(), pivot.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5, y.data_ptr(), BF,
                           mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), None, None, 0.0, B, P, C, G, 1, st)
    def f_small():
        L.norm_small_fwd(x.data_ptr(), None, 0, None, gamma.data_ptr(), beta.data_ptr(), 1e-5, y.data_ptr(), mean.data_ptr(),
                         rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), B, P, C, G, 1, st)
    def b_chain():
        L.norm_bwd_reduce(dA.data_ptr(), BF, x.data_ptr(), BF, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                          sums2.data_ptr(), B, P, C, G, 1, nrep, st)
        L.norm_bwd_apply_fused_bias(dA.data_ptr(), BF, x.data_ptr(), BF, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                    rstd.data_ptr(), gamma.data_ptr(), sums2.data_ptr(), dx.data_ptr(), BF, dg.data_ptr(), db.data_ptr(),
                                    sums.data_ptr(), pivot.data_ptr(), dbias.data_ptr(), B, P, C, G, 1, nrep, st)
    def b_small():
        L.norm_small_bwd(dA.data_ptr(), x.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                         gamma.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), dbias.data_ptr(), B, P, C, G, 1, st)
    f_chain()
    mb = x.numel() * 2 / 1e6
    print("H=%3d C=%3d %6.1f MB | fwd chain %7.1f us  small %7.1f us | bwd chain %7.1f us  small %7.1f us" %
          (H, C, mb, timeit(f_chain), timeit(f_small), timeit(b_chain), timeit(b_small)), flush=True)
