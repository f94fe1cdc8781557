"""Micro-benchmark + cross-check (GPU box): k_conv3x3_c32 (PHX_C32=1: filter in registers, persistent, double-staged patches) against
k_conv3x3_fwd_dma128<32> (PHX_C32=0) on the 32 -> 32-channel layers, through the same ABI entry point and buffers."""
import os, sys
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
for (B, H, W) in [(64, 128, 128), (64, 64, 64), (7, 48, 96), (64, 128, 128)]:
    K = N = 32
    x = torch.relu(torch.randn(B, H, W, K, device="cuda")).to(torch.bfloat16)
    wf = (torch.randn(9 * K * N, device="cuda") * 0.05).to(torch.bfloat16)
    res, row = {}, []
    for mode in ("0", "1"):
        os.environ["PHX_C32"] = mode
        os.environ["PHX_FWD_WS"] = "5"
        y = torch.zeros(B, H, W, N, device="cuda", dtype=torch.bfloat16)
        ntile = L.conv3x3_mfma_bf16_tiles(B, H, W, K, N)
        part = torch.zeros(ntile * 2 * N, device="cuda")
        def run():
            L.conv3x3_mfma_bf16(x.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, part.data_ptr(), B, H, W, K, N, st)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        res[mode] = (y.float().clone(), part.view(ntile, 2, N).sum(0).clone())
        gb = 2.0 * B * H * W * (K + N) / 1e9
        row.append("C32=%s %7.1f us %6.1f TF %5.2f TB/s" % (mode, ms * 1e3, 18.0 * K * N * B * H * W / ms / 1e9, gb / ms))
    dy = (res["0"][0] - res["1"][0]).abs().max().item()
    ds = ((res["0"][1] - res["1"][1]).abs() / (res["0"][1].abs() + 1.0)).max().item()
    print("%-16s %s | max|dy| %.3g stats rel %.3g" % ((B, H, W), "  |  ".join(row), dy, ds), flush=True)
