"""dev helper (GPU box): cycle stamps of block 0 (all eight waves) of k_conv3x3_pp over its first work items."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.debug_lib()
st = torch.cuda.current_stream().cuda_stream
L.debug_conv_policy(2, 1)
B, H, W, K, N = [int(v) for v in (sys.argv[1:6] if len(sys.argv) >= 6 else (64, 128, 128, 128, 128))]
stats = len(sys.argv) > 6 and sys.argv[6] == "stats"
x = torch.relu(torch.randn(B, H, W, K, device="cuda")).to(torch.bfloat16)
wf = (torch.randn(9 * K * N, device="cuda") * 0.05).to(torch.bfloat16)
y = torch.zeros(B, H, W, N, device="cuda", dtype=torch.bfloat16)
part = torch.zeros(L.conv3x3_mfma_bf16_tiles(B, H, W, K, N) * 2 * N, device="cuda")
tr = torch.zeros(1024 * 8, dtype=torch.int64, device="cuda")
def run():
    L.conv3x3_mfma_bf16(x.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, part.data_ptr() if stats else None, B, H, W, K, N, st)
for _ in range(3): run()
torch.cuda.synchronize()
L.debug_set_trace(tr.data_ptr())
run()
torch.cuda.synchronize()
L.debug_set_trace(None)
t = tr.cpu().numpy().reshape(1024, 8).astype(np.int64)
nch = K // 32
names = ["compute", "->bar", "patchDMA", "epilogue", "slabDMA", "vm wait", "barrier"]
for wv in (0, 4):
    print("wave %d (%s)" % (wv, "A" if wv < 4 else "B"))
    for it in range(0, min(3 * nch, 11)):
        r = t[it * 8: it * 8 + 8, wv]
        if r[0] == 0: break
        nxt = t[(it + 1) * 8, wv]
        d = [r[k + 1] - r[k] for k in range(7)]
        print("  item %2d: " % it + "  ".join("%s %5d" % (n, v) for n, v in zip(names, d)) + "   | body %6d" % (nxt - r[0] if nxt else 0))
# spread over the waves of a half at the end of compute
it = nch
print("end-of-compute stamps, item %d, waves 0-7 relative to wave 0:" % it, (t[it * 8 + 1] - t[it * 8 + 1, 0]).tolist())
print("start-of-compute stamps:", (t[it * 8 + 0] - t[it * 8 + 0, 0]).tolist())
e = t[88:96]
for wv in (0, 4):
    d = e[1:, wv] - e[:-1, wv]
    print("last epilogue, wave %d: W0 %d  R0 %d  W1 %d  S0 %d  R1+W2 %d  (S1 R2 W3 S2 R3) %d  S3 %d   total %d" % ((wv,) + tuple(d.tolist()) + (e[7, wv] - e[0, wv],)))
