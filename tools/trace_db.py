"""dev helper (GPU box): cycle stamps of block 0 / wave 0 of k_conv3x3_fwd_db over its first chunks."""
import os, sys
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
os.environ["PHX_FWD_WS"] = "5"; os.environ["PHX_FWD_DB"] = "1"
B, H, W, K, N = [int(v) for v in (sys.argv[1:6] if len(sys.argv) >= 6 else (64, 128, 128, 128, 128))]
x = torch.relu(torch.randn(B, H, W, K, device="cuda")).to(torch.bfloat16)
wf = (torch.randn(9 * K * N, device="cuda") * 0.05).to(torch.bfloat16)
y = torch.zeros(B, H, W, N, device="cuda", dtype=torch.bfloat16)
part = torch.zeros(L.conv3x3_mfma_bf16_tiles(B, H, W, K, N) * 2 * N, device="cuda")
tr = torch.zeros(4096, dtype=torch.int64, device="cuda")
for ab in sys.argv[6:] or [""]:
    os.environ.pop("PHX_DBG_ABLATE", None)
    if ab: os.environ["PHX_DBG_ABLATE"] = ab
    for _ in range(3):
        L.conv3x3_mfma_bf16(x.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, part.data_ptr(), B, H, W, K, N, st)
    torch.cuda.synchronize()
    L.debug_set_trace(tr.data_ptr())
    L.conv3x3_mfma_bf16(x.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, part.data_ptr(), B, H, W, K, N, st)
    torch.cuda.synchronize()
    L.debug_set_trace(None)
    t = tr.cpu().numpy()
    print("ablate=%s  prologue %d cycles" % (ab or "-", t[3] - t[0]))
    for s in range(0, 14):
        a, b, c, d = t[4 * s + 3], t[4 * s + 4], t[4 * s + 5], t[4 * (s + 1) + 3]
        print("  chunk %2d: wait+barrier %6d  epilogue/set_item %6d  compute %6d" % (s, b - a, c - b, d - c))
