"""dev (GPU box): phase stamps (shader clock) of block 0 of the small-map forward / filter-gradient kernels."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
for (B, H, W, K, N) in [(64, 2, 2, 192, 192), (64, 4, 4, 192, 192), (64, 8, 8, 192, 192), (64, 16, 16, 192, 192)]:
    x = torch.randn(B, H, W, K, device="cuda").to(torch.bfloat16)
    dy = torch.randn(B, H, W, N, device="cuda").to(torch.bfloat16)
    wf = torch.randn(9 * K * N, device="cuda").to(torch.bfloat16)
    y = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16)
    dw = torch.zeros(9 * K * N, device="cuda")
    fwsb = int(L.conv3x3_mfma_ws_bytes(B, H, W, K, N)); fws = torch.empty(max(fwsb // 4, 1), device="cuda")
    wsb = int(L.conv3x3_wgrad_ws_bytes(B, H, W, K, N)); ws = torch.empty(max(wsb // 4, 1), device="cuda")
    tr = torch.zeros(16, dtype=torch.int64, device="cuda")
    for which in ("fwd", "wgrad"):
        def run():
            if which == "fwd":
                L.conv3x3_mfma_bf16_ws(x.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, None, fws.data_ptr() if fwsb else None, fwsb, B, H, W, K, N, st)
            else:
                L.conv3x3_wgrad_mfma_bf16(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), wsb, B, H, W, K, N, st)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        tr.zero_(); L.debug_set_trace(tr.data_ptr()); run(); torch.cuda.synchronize(); L.debug_set_trace(None)
        t = tr.cpu().numpy()
        nz = [(i, int(v - t[0])) for i, v in enumerate(t) if v]
        print("%s %-22s %6.1f us/launch (back to back); block-0 stamps (slot, cycles from start): %s" % (which, (B, H, W, K, N), us, nz))
