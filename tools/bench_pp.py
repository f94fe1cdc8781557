"""GPU box: the large-map kernels (k_conv3x3_pp, the anti-phase pair kernel; k_conv3x3_c32 for 32 -> 32) against the 256-pixel
kernel k_conv3x3_mfma on the large-map shapes of phiseg_7_5 at batch 64, each launch alone (HIP events).
usage: python tools/bench_pp.py [shape=i,j,...]"""
import os, sys
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.debug_lib()
st = torch.cuda.current_stream().cuda_stream
shapes = [(64, 128, 128, 128, 128), (64, 64, 64, 192, 192), (64, 128, 128, 64, 128), (64, 64, 64, 128, 192), (64, 128, 128, 192, 32),
          (64, 128, 128, 64, 64), (64, 64, 64, 64, 64), (64, 128, 128, 32, 32), (64, 128, 128, 32, 64), (64, 32, 32, 128, 128)]
abl = len(sys.argv) > 1 and sys.argv[1] == "ablate"
for a in sys.argv[1:]:
    if a.startswith("shape="):
        shapes = [shapes[int(v)] for v in a[6:].split(",")]
for (B, H, W, K, N) in shapes:
    x = torch.relu(torch.randn(B, H, W, K, device="cuda")).to(torch.bfloat16)
    wf = (torch.randn(9 * K * N, device="cuda") * 0.05).to(torch.bfloat16)
    y = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16)
    fl = 18.0 * K * N * B * H * W
    row = []
    modes = [("k256", 0, False), ("large", 2, False), ("large+stats", 2, True), ("k256+stats", 0, True)]
    for name, pol, stats in modes:
        L.debug_conv_policy(pol, 1)
        part = torch.zeros(L.conv3x3_mfma_bf16_tiles(B, H, W, K, N) * 2 * N, device="cuda", dtype=torch.float32)
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
