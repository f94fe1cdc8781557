"""Debug (GPU box): per-block start/end/placement log of the MFMA conv kernels -> concurrency, clock, round structure."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
shapes = [(64, 128, 128, 128, 128), (64, 64, 64, 192, 192), (64, 32, 32, 192, 192), (64, 16, 16, 192, 192)]
for (B, H, W, K, N) in shapes:
    x = torch.randn(B, H, W, K, device="cuda").to(torch.bfloat16)
    dy = torch.randn(B, H, W, N, device="cuda").to(torch.bfloat16)
    wf = torch.randn(9 * K * N, device="cuda").to(torch.bfloat16)
    y = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16)
    dw = torch.zeros(9 * K * N, device="cuda")
    wsb = int(L.conv3x3_wgrad_ws_bytes(B, H, W, K, N))
    ws = torch.empty(max(wsb // 4, 1), device="cuda")
    def run():
        if which == "wgrad":
            L.conv3x3_wgrad_mfma_bf16(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), wsb, B, H, W, K, N, st)
        else:
            L.conv3x3_mfma_bf16(x.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, None, B, H, W, K, N, st)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    log = torch.zeros(4 * 65536, dtype=torch.int64, device="cuda")
    L.debug_set_blocklog(log.data_ptr())
    run(); torch.cuda.synchronize()
    L.debug_set_blocklog(None)
    a = log.cpu().numpy().reshape(-1, 4)
    a = a[a[:, 1] != 0]
    nb = len(a)
    if nb == 0:
        print(which, (B, H, W, K, N), 'kernel %.1f us: no block log (pair kernel)' % us)
        continue
    os.makedirs('gpurun_out', exist_ok=True); np.save('gpurun_out/blocklog_%s_%d_%d.npy' % (which, H, K), a)
    # the shader-clock counter is per XCD (unsynchronised): cluster the blocks by counter epoch
    order = np.argsort(a[:, 0])
    a = a[order]
    xcc = (a[:, 2] >> 32) & 0xf
    groups = [np.where(xcc == v)[0] for v in np.unique(xcc)]
    hw = a[:, 2] & 0xffffffff
    rows = []
    for gi in groups:
        g = a[gi]
        span = g[:, 1].max() - g[:, 0].min()
        rspan = (g[:, 3].max() - g[:, 3].min()) * 10.0          # ns (100 MHz realtime counter)
        dur = (g[:, 1] - g[:, 0]).astype(np.float64)
        ncu = len(np.unique((hw[gi] >> 8) & 0xff))
        rows.append((len(gi), span, span / max(rspan, 1), dur.mean(), dur.min(), dur.max(), ncu, dur.sum() / span / max(ncu, 1)))
    r = np.array(rows)
    print("%s %s: kernel %.1f us; blocks %d in %d XCD groups; per group: blocks %.0f, span %.0f ticks, %.2f ticks/ns, block dur mean %.0f "
          "(min %.0f max %.0f), CUs %.0f, mean resident blocks/CU %.2f" % (which, (B, H, W, K, N), us, nb, len(groups), r[:, 0].mean(),
          r[:, 1].mean(), r[:, 2].mean(), r[:, 3].mean(), r[:, 4].min(), r[:, 5].max(), r[:, 6].mean(), r[:, 7].mean()))
