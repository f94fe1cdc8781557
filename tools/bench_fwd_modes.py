"""Micro-benchmark (GPU box): forward / data-gradient 3x3 MFMA convolution, default policy vs the experimental tile designs
(PHX_FWD_WS = 2: 8 MFMA waves x 64 px; 3: 4 MFMA waves x 128 px, shared patch rows), same process, same buffers."""
import os, sys
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
shapes = [(64, 128, 128, 128, 128), (64, 128, 128, 64, 128), (64, 128, 128, 128, 64), (64, 128, 128, 32, 192), (64, 64, 64, 192, 192),
          (64, 64, 64, 64, 64), (64, 64, 64, 128, 192), (64, 64, 64, 192, 64), (64, 32, 32, 128, 128), (64, 32, 32, 192, 192),
          (64, 32, 32, 64, 64), (64, 32, 32, 256, 192)]
modes = sys.argv[1:] or ["0", "2", "3"]
if os.environ.get("BENCH_SHAPES") == "n32":
    shapes = [(64, 128, 128, 32, 32), (64, 128, 128, 192, 32), (64, 128, 128, 64, 32), (64, 64, 64, 32, 32), (64, 128, 128, 32, 64)]
if os.environ.get("BENCH_SHAPES") == "short":
    shapes = [(64, 128, 128, 128, 128), (64, 64, 64, 192, 192), (64, 128, 128, 32, 192), (64, 64, 64, 64, 64)]
for (B, H, W, K, N) in shapes:
    data = os.environ.get("BENCH_DATA", "randn")
    if data == "zeros":
        x = torch.zeros(B, H, W, K, device="cuda", dtype=torch.bfloat16)
        wf = torch.zeros(9 * K * N, device="cuda", dtype=torch.bfloat16)
    elif data == "relu":
        x = torch.relu(torch.randn(B, H, W, K, device="cuda")).to(torch.bfloat16)
        wf = (torch.randn(9 * K * N, device="cuda") * 0.05).to(torch.bfloat16)
    else:
        x = torch.randn(B, H, W, K, device="cuda").to(torch.bfloat16)
        wf = (torch.randn(9 * K * N, device="cuda") * 0.05).to(torch.bfloat16)
    y = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16)
    row = []
    for m in modes:
        os.environ["PHX_FWD_PP"] = "2" if m == "pp" else "0"
        os.environ["PHX_FWD_WS"] = "0" if m == "pp" else m
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
        row.append("ws=%s %7.3f ms %7.1f TF" % (m, ms, 18.0 * K * N * B * H * W / ms / 1e9))
    print("%-26s %s" % ((B, H, W, K, N), "  |  ".join(row)), flush=True)
