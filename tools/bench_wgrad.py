"""Micro-benchmark (GPU box): bf16 MFMA filter-gradient (+ reduction) / forward launches on chosen shapes, back to back on one stream."""
import ctypes, sys, os
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
shapes = [(64, 2, 2, 192, 192), (64, 4, 4, 192, 192), (64, 8, 8, 192, 192), (64, 16, 16, 192, 192), (64, 128, 128, 32, 32), (64, 128, 128, 128, 128), (64, 128, 128, 192, 32), (64, 64, 64, 192, 192),
          (64, 32, 32, 128, 128), (64, 32, 32, 192, 192), (64, 16, 16, 192, 192), (64, 8, 8, 192, 192), (64, 4, 4, 192, 192)]
which = sys.argv[1] if len(sys.argv) > 1 else "wgrad"
for (B, H, W, K, N) in shapes:
    x = torch.randn(B, H, W, K, device="cuda").to(torch.bfloat16)
    dy = torch.randn(B, H, W, N, device="cuda").to(torch.bfloat16)
    dw = torch.zeros(9 * K * N, device="cuda")
    wsb = int(L.conv3x3_wgrad_ws_bytes(B, H, W, K, N)) if os.environ.get("PHX_WS", "1") == "1" else 0
    ws = torch.empty(max(wsb // 4, 1), device="cuda")
    wsp = ws.data_ptr() if wsb else None
    wf = torch.randn(9 * K * N, device="cuda").to(torch.bfloat16)
    y = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16)
    fwsb = int(L.conv3x3_mfma_ws_bytes(B, H, W, K, N)) if os.environ.get("PHX_FWD_WS", "1") == "1" else 0
    fws = torch.empty(max(fwsb // 4, 1), device="cuda")
    def run():
        if which == "wgrad":
            L.conv3x3_wgrad_mfma_bf16(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), wsp, wsb, B, H, W, K, N, st)
        else:
            L.conv3x3_mfma_bf16_ws(x.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, None, fws.data_ptr() if fwsb else None, fwsb,
                                   B, H, W, K, N, st)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%s %-26s %8.3f ms %8.1f TFLOP/s" % (which, (B, H, W, K, N), ms, 18.0 * K * N * B * H * W / ms / 1e9))
