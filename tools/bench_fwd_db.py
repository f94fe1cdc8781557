"""Micro-benchmark + cross-check (GPU box): k_conv3x3_fwd_db (PHX_FWD_DB=1, double-buffered persistent) against
k_conv3x3_fwd_dma128 (PHX_FWD_DB=0) through the same ABI entry point, same process, same buffers.
usage: python tools/bench_fwd_db.py [ablate-bits ...]   (PHX_DBG_ABLATE values to time besides the full kernel)"""
import os, sys
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
shapes = [(64, 128, 128, 128, 128), (64, 128, 128, 64, 128), (64, 128, 128, 128, 64), (64, 128, 128, 32, 192), (64, 64, 64, 192, 192),
          (64, 64, 64, 64, 64), (64, 64, 64, 128, 192), (64, 64, 64, 192, 64), (64, 128, 128, 32, 64), (64, 64, 64, 32, 64), (8, 32, 32, 128, 128),
          (3, 16, 32, 64, 64), (64, 32, 32, 192, 192)]
if os.environ.get("BENCH_SHAPES") == "short":
    shapes = [(64, 128, 128, 128, 128), (64, 64, 64, 192, 192), (64, 128, 128, 32, 192), (64, 64, 64, 64, 64)]
abl = sys.argv[1:]
os.environ["PHX_FWD_WS"] = "5"          # the 16 x 32-tile kernels whenever the shape is eligible
for (B, H, W, K, N) in shapes:
    x = torch.relu(torch.randn(B, H, W, K, device="cuda")).to(torch.bfloat16)
    wf = (torch.randn(9 * K * N, device="cuda") * 0.05).to(torch.bfloat16)
    res = {}
    row = []
    for mode in ["0", "1"] + ["1:" + a for a in abl] + (["1:dps2", "1:dps3"] if os.environ.get("BENCH_DPS") else []):
        db, _, ab = mode.partition(":")
        os.environ["PHX_FWD_DB"] = db
        os.environ.pop("PHX_DBG_ABLATE", None); os.environ.pop("PHX_DB_DPS", None)
        if ab.startswith("dps"): os.environ["PHX_DB_DPS"] = ab[3:]
        elif ab: os.environ["PHX_DBG_ABLATE"] = ab
        y = torch.zeros(B, H, W, N, device="cuda", dtype=torch.bfloat16)
        ntile = L.conv3x3_mfma_bf16_tiles(B, H, W, K, N)
        part = torch.zeros(ntile * 2 * N, device="cuda")
        def run():
            L.conv3x3_mfma_bf16(x.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, part.data_ptr(), B, H, W, K, N, st)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        if not ab:
            res[db] = (y.float().clone(), part.view(ntile, 2, N).sum(0).clone())
        row.append("%s %7.3f ms %7.1f TF" % (mode, ms, 18.0 * K * N * B * H * W / ms / 1e9))
    dy = (res["0"][0] - res["1"][0]).abs().max().item()
    ds = ((res["0"][1] - res["1"][1]).abs() / (res["0"][1].abs() + 1.0)).max().item()
    print("%-26s %s  | max|dy| %.3g  stats rel %.3g" % ((B, H, W, K, N), "  |  ".join(row), dy, ds), flush=True)
