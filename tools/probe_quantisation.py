"""GPU box: launch time of the 256-pixel kernel on the H = 16 shape 192 -> 192 against the batch size (each launch alone, HIP events):
the work is B x 24 wave tiles of 64 pixels x 32 channels (216 MFMAs each) on 1 024 SIMDs -- the time should step where the tile count
crosses a multiple of 1 024 (B = 42.7, 85.3, 128), not grow with it.  Round-5 evidence for DESIGN.md section 5 (re-blocking of the
H = 16 / 32 kernels).  usage: python tools/probe_quantisation.py"""
import sys
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
for (H, K, N) in [(16, 192, 192), (32, 128, 128)]:
    for B in (8, 16, 24, 32, 40, 42, 44, 48, 56, 64, 80, 84, 88, 96, 112, 128):
        x = torch.randn(B, H, H, K, device="cuda").to(torch.bfloat16)
        wf = (torch.randn(9 * K * N, device="cuda") * 0.05).to(torch.bfloat16)
        y = torch.empty(B, H, H, N, device="cuda", dtype=torch.bfloat16)
        wsb = int(L.conv3x3_mfma_ws_bytes(B, H, H, K, N))
        ws = torch.empty(max(wsb // 4, 1), device="cuda")
        run = lambda: L.conv3x3_mfma_bf16_ws(x.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, None, ws.data_ptr() if wsb else None, wsb, B, H, H, K, N, st)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / 20
        tiles = B * (H // 16) ** 2
        blk = 32 if tiles * (N // 64) <= 256 else 64
        wt = tiles * 4 * (N // blk)
        print("%3d -> %3d @ %2d x %2d  B = %3d  %5d wave tiles of 64 px x %d ch (%.2f per SIMD, split-K %d)  %7.1f us  %6.0f TFLOP/s"
              % (K, N, H, H, B, wt, blk, wt / 1024.0, int(L.conv3x3_mfma_ksplit(B, H, H, K, N)), us, 18.0 * K * N * B * H * H / us / 1e6), flush=True)
