"""GPU box: k_conv3x3_pp (anti-phase pair kernel) against k_conv3x3_fwd_dma128 on the large-map shapes of phiseg_7_5 at batch 64.
usage: python tools/bench_pp.py [ablate]      (ablate: also the PHX_DBG_ABLATE masks 1 no patch DMA, 2 no slab DMA, 4 no MFMAs, 8 no epilogue)"""
import os, sys
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
os.environ["PHX_FWD_WS"] = "5"
os.environ["PHX_FWD_DB"] = "0"
shapes = [(64, 128, 128, 128, 128), (64, 64, 64, 192, 192), (64, 128, 128, 64, 128), (64, 64, 64, 128, 192), (64, 128, 128, 192, 32),
          (64, 128, 128, 64, 64), (64, 64, 64, 64, 64), (64, 128, 128, 32, 32), (64, 128, 128, 32, 64), (64, 32, 32, 128, 128)]
abl = len(sys.argv) > 1 and sys.argv[1] == "ablate"
for (B, H, W, K, N) in shapes:
    x = torch.relu(torch.randn(B, H, W, K, device="cuda")).to(torch.bfloat16)
    wf = (torch.randn(9 * K * N, device="cuda") * 0.05).to(torch.bfloat16)
    y = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16)
    part = torch.zeros(B * (H // 16) * (W // 32) * 2 * N, device="cuda", dtype=torch.float32)
    fl = 18.0 * K * N * B * H * W
    row = []
    modes = [("dma128", "0", "0", False), ("pp", "1", "0", False), ("pp+stats", "1", "0", True), ("dma128+stats", "0", "0", True)]
    if abl:
        modes += [("pp/%s" % m, "1", m, False) for m in ("1", "2", "3", "4", "16", "32")]
    for name, pp, m, stats in modes:
        os.environ["PHX_FWD_PP"] = "2" if pp == "1" else "0"
        os.environ["PHX_DBG_ABLATE"] = m
        def run():
            L.conv3x3_mfma_bf16(x.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, part.data_ptr() if stats else None, B, H, W, K, N, st)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        row.append("%s %.3f ms %4.0f TF" % (name, ms, fl / ms / 1e9))
    print("%-24s %s" % ((B, H, W, K, N), " | ".join(row)), flush=True)
