"""Debug (GPU box): phase timestamps (ns) inside k_conv3x3_mfma for a few shapes (block 0, thread 0)."""
import sys
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
tr = torch.zeros(16, dtype=torch.int64, device="cuda")
for (B, H, W, K, N) in [(64, 128, 128, 128, 128), (64, 64, 64, 192, 192), (64, 16, 16, 192, 192), (64, 4, 4, 192, 192)]:
    x = torch.randn(B, H, W, K, device="cuda").to(torch.bfloat16)
    wf = torch.randn(9 * K * N, device="cuda").to(torch.bfloat16)
    y = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(2):
        L.conv3x3_mfma_bf16(x.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, None, B, H, W, K, N, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        L.conv3x3_mfma_bf16(x.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, None, B, H, W, K, N, st)
    e1.record(); torch.cuda.synchronize()
    L.debug_set_trace(tr.data_ptr())
    L.conv3x3_mfma_bf16(x.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, None, B, H, W, K, N, st)
    torch.cuda.synchronize()
    L.debug_set_trace(None)
    t = tr.cpu().tolist()
    names = ["start(plan done)", "prefetch0 issued", "chunk1: sync1", "chunk1: LDS staged", "chunk1: prefetch issued", "all chunks done", "outputs stored"]
    print((B, H, W, K, N), "kernel %.1f us" % (e0.elapsed_time(e1) * 200), {names[i]: t[i] - t[0] for i in range(1, 7)},
          "| epilogue: sync", t[7] - t[5], "lds writes", t[8] - t[7], "sync", t[9] - t[8], "global stores", t[6] - t[9])
