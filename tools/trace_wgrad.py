"""Debug (GPU box): phase timestamps inside k_conv3x3_wgrad for one shape."""
import sys, os
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
tr = torch.zeros(16, dtype=torch.int64, device="cuda")
for (B, H, W, K, N) in [(64, 32, 32, 128, 128), (64, 128, 128, 128, 128), (64, 128, 128, 32, 32)]:
    x = torch.randn(B, H, W, K, device="cuda").to(torch.bfloat16)
    dy = torch.randn(B, H, W, N, device="cuda").to(torch.bfloat16)
    dw = torch.zeros(9 * K * N, device="cuda")
    wsb = int(L.conv3x3_wgrad_ws_bytes(B, H, W, K, N)) if os.environ.get("PHX_WS", "1") == "1" else 0
    ws = torch.empty(max(wsb // 4, 1), device="cuda")
    wsp = ws.data_ptr() if wsb else None
    for _ in range(2):
        L.conv3x3_wgrad_mfma_bf16(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), wsp, wsb, B, H, W, K, N, st)
    torch.cuda.synchronize()
    L.debug_set_trace(tr.data_ptr())
    L.conv3x3_wgrad_mfma_bf16(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), wsp, wsb, B, H, W, K, N, st)
    torch.cuda.synchronize()
    L.debug_set_trace(None)
    t = tr.cpu().tolist()
    names = ["start", "prefetch0 issued", "first sync", "LDS staged", "prefetch1 issued", "compute done (all tiles)", "atomics issued"]
    print((B, H, W, K, N), "cycles since start:", {names[i]: t[i] - t[0] for i in range(1, 7)})
