"""Micro-benchmark (GPU box): forward 3x3 MFMA convolution on the small maps (H <= 16), launches back to back on one stream,
with the split-K finish included, against the floor of a trivial dependent launch."""
import os, sys
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
shapes = [(64, 32, 32, 128, 128), (64, 32, 32, 192, 192), (64, 16, 16, 192, 192), (64, 16, 16, 384, 192), (64, 8, 8, 192, 192),
          (64, 4, 4, 192, 192), (64, 2, 2, 192, 192), (64, 8, 8, 384, 192), (64, 4, 4, 384, 192)]
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
t = torch.zeros(256, device="cuda")
print("floor: add_ of 256 floats %.2f us" % timeit(lambda: t.add_(1.0)))
for (B, H, W, K, N) in shapes:
    x = torch.randn(B, H, W, K, device="cuda").to(torch.bfloat16)
    wf = (torch.randn(9 * K * N, device="cuda") * 0.05).to(torch.bfloat16)
    y = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16)
    ntile = L.conv3x3_mfma_bf16_tiles(B, H, W, K, N)
    part = torch.zeros(max(ntile, 1) * 2 * N, device="cuda")
    row = []
    for env in sys.argv[1:] or [""]:
        for kv in env.split(","):
            if kv: os.environ[kv.split("=")[0]] = kv.split("=")[1]
        wsb = int(L.conv3x3_mfma_ws_bytes(B, H, W, K, N))
        ws = torch.empty(max(wsb // 4, 1), device="cuda")
        ks = L.conv3x3_mfma_ksplit(B, H, W, K, N)
        def run():
            L.conv3x3_mfma_bf16_ws(x.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, None if wsb else part.data_ptr(), ws.data_ptr() if wsb else None, wsb, B, H, W, K, N, st)
        us = timeit(run)
        row.append("%-18s ksplit %2d %6.2f us %6.1f TF" % (env or "default", ks, us, 18.0 * K * N * B * H * W / us / 1e6))
        for kv in env.split(","):
            if kv: os.environ.pop(kv.split("=")[0])
    print("%-24s %s" % ((B, H, W, K, N), " | ".join(row)), flush=True)
