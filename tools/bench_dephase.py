import os, sys
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
os.environ["PHX_FWD_WS"] = "5"
shapes = [(64, 128, 128, 128, 128), (64, 128, 128, 64, 128), (64, 64, 64, 192, 192), (64, 64, 64, 64, 64), (64, 32, 32, 128, 128)]
for (B, H, W, K, N) in shapes:
    x = torch.randn(B, H, W, K, device="cuda").to(torch.bfloat16)
    wf = (torch.randn(9 * K * N, device="cuda") * 0.05).to(torch.bfloat16)
    y = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16)
    row = []
    for m in ["0", "10", "20", "30", "40", "60", "80", "120"]:
        os.environ["PHX_DEPHASE"] = m
        def run():
            L.conv3x3_mfma_bf16(x.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, None, B, H, W, K, N, st)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        row.append("%s:%6.3f(%4.0f)" % (m, ms, 18.0 * K * N * B * H * W / ms / 1e9))
    print("%-26s ms(TF) by dephase  %s" % ((B, H, W, K, N), "  ".join(row)), flush=True)
