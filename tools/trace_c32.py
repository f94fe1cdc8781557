"""dev (GPU box): phase stamps (shader clock) of block 0 / wave 0 of k_conv3x3_c32 per tile iteration:
slots 0 top, 1 patch landed + barrier, 2 next patch issued, 3 MFMAs issued, 4 all waves done, 5 tile packed to LDS, 6 stores issued"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
B, H, W, K, N = 64, 128, 128, 32, 32
x = torch.randn(B, H, W, K, device="cuda").to(torch.bfloat16)
wf = torch.randn(9 * K * N, device="cuda").to(torch.bfloat16)
y = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16)
ntile = L.conv3x3_mfma_bf16_tiles(B, H, W, K, N)
part = torch.zeros(ntile * 2 * N, device="cuda")
for stats in (True, False):
    def run():
        L.conv3x3_mfma_bf16(x.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, part.data_ptr() if stats else None, B, H, W, K, N, st)
    for _ in range(3): run()
    tr = torch.zeros(512, dtype=torch.int64, device="cuda")
    L.debug_set_trace(tr.data_ptr()); run(); torch.cuda.synchronize(); L.debug_set_trace(None)
    t = tr.cpu().numpy().reshape(-1, 8)
    t0 = t[0, 0]
    print("stats" if stats else "no stats")
    for it in range(8):
        if t[it, 0]:
            r = t[it, :7] - t0
            print("  tile %d: top %6d | wait+barrier %5d | issue %5d | mfma %5d | barrier %5d | pack %5d | read+store %5d" %
                  (it, r[0], r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], r[6] - r[5]))
