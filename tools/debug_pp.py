"""dev (GPU box): where does k_conv3x3_pp differ from the 256-pixel kernel?  usage: B H W K N [affine|bias|plain] [relu]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
B, H, W, K, N = [int(v) for v in sys.argv[1:6]]
mode = sys.argv[6] if len(sys.argv) > 6 else "plain"
act = 1 if "relu" in sys.argv else 0
torch.manual_seed(0)
x = torch.randn(B, H, W, K, device="cuda").to(torch.bfloat16)
wf = (torch.randn(9 * K * N, device="cuda") * 0.05).to(torch.bfloat16)
scale = (1 + 0.2 * torch.randn(N, device="cuda")).float()
shift = (0.1 * torch.randn(N, device="cuda")).float()
def run(env):
    for k, v in env.items(): os.environ[k] = v
    y = torch.full((B, H, W, N), 7.0, device="cuda", dtype=torch.bfloat16)
    if mode == "affine":
        L.conv3x3_mfma_bf16_affine(x.data_ptr(), wf.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr(), act, None, 0, B, H, W, K, N, st)
    elif mode == "bias":
        L.conv3x3_mfma_bf16(x.data_ptr(), wf.data_ptr(), y.data_ptr(), shift.data_ptr(), act, None, B, H, W, K, N, st)
    else:
        L.conv3x3_mfma_bf16(x.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, None, B, H, W, K, N, st)
    torch.cuda.synchronize()
    return y.float().cpu().numpy()
ref = run({"PHX_FWD_WS": "0"})
for trial in range(3):
    got = run({"PHX_FWD_WS": "5", "PHX_FWD_PP": "2"})
    bad = ~np.isclose(got, ref, rtol=2e-2, atol=2e-2)
    print("trial", trial, "bad", bad.sum(), "nan", np.isnan(got).sum())
    if bad.any():
        b, h, w, n = np.nonzero(bad)
        print("  images", np.unique(b), "rows", np.unique(h), "cols", np.unique(w)[:40], "chan", np.unique(n)[:70])
        print("  sample got/ref", got[bad][:8], ref[bad][:8])
