"""GPU box: k_conv3x3_fwd_dma128 with parts switched off (PHX_DBG_ABLATE bits: 1 no patch loads, 2 no slab loads, 4 no MFMAs)."""
import os, sys
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
mode = sys.argv[1] if len(sys.argv) > 1 else "5"          # "db": k_conv3x3_fwd_db (bits as above, 8 no output stores, 16 no DMA at all)
os.environ["PHX_FWD_WS"] = "5"
os.environ["PHX_FWD_DB"] = "1" if mode == "db" else "0"
shapes = [(64, 128, 128, 128, 128), (64, 64, 64, 192, 192), (64, 64, 64, 64, 64), (64, 32, 32, 128, 128)]
for (B, H, W, K, N) in shapes:
    x = torch.randn(B, H, W, K, device="cuda").to(torch.bfloat16)
    wf = (torch.randn(9 * K * N, device="cuda") * 0.05).to(torch.bfloat16)
    y = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16)
    row = []
    for m in (["0", "1", "2", "3", "4", "7", "8", "16", "20"] if mode == "db" else ["0", "1", "2", "3", "4"]):
        os.environ["PHX_DBG_ABLATE"] = m
        def run():
            L.conv3x3_mfma_bf16(x.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, None, B, H, W, K, N, st)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        row.append("%s:%6.3f" % (m, e0.elapsed_time(e1) / 10))
    print("%-26s ms by ablation mask  %s" % ((B, H, W, K, N), "  ".join(row)), flush=True)
