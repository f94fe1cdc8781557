"""GPU box: the 32 -> 32 @ 128 x 128 layer with the producer's scale-shift-ReLU re-formed inside the kernels (phx_conv3x3_mfma_bf16_xf,
phx_conv3x3_wgrad_mfma_bf16_partial_xf) against apply pass + plain kernels, each launch alone (HIP events), batch 64."""
import sys
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
BF = rt.BF16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


B, H, W, K, N = 64, 128, 128, 32, 32
x = torch.randn(B, H, W, K, device="cuda").to(torch.bfloat16)
sc = (1.0 + 0.2 * torch.randn(K, device="cuda")).float().contiguous()
sh = (0.1 * torch.randn(K, device="cuda")).float().contiguous()
a = torch.empty_like(x)
wf = (torch.randn(9 * K * N, device="cuda") * 0.05).to(torch.bfloat16)
y1, y2 = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16), torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16)
P = B * H * W
nt = L.conv3x3_mfma_bf16_tiles(B, H, W, K, N)
part = torch.zeros(nt * 2 * N, device="cuda")
dy = torch.randn(B, H, W, N, device="cuda").to(torch.bfloat16)
wsb = int(L.conv3x3_wgrad_ws_bytes(B, H, W, K, N))
ws = torch.empty(wsb // 4, device="cuda")
dw = torch.zeros(9 * K * N, device="cuda")
ap = lambda: L.affine_act(x.data_ptr(), BF, sc.data_ptr(), sh.data_ptr(), a.data_ptr(), BF, 1, P, K, 1, st)
cv = lambda: L.conv3x3_mfma_bf16(a.data_ptr(), wf.data_ptr(), y1.data_ptr(), None, 0, part.data_ptr(), B, H, W, K, N, st)
cx = lambda: L.conv3x3_mfma_bf16_xf(x.data_ptr(), sc.data_ptr(), sh.data_ptr(), wf.data_ptr(), y2.data_ptr(), part.data_ptr(), B, H, W, K, N, st)
wg = lambda: L.conv3x3_wgrad_mfma_bf16_partial(a.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), wsb, B, H, W, K, N, st)
wx = lambda: L.conv3x3_wgrad_mfma_bf16_partial_xf(x.data_ptr(), sc.data_ptr(), sh.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), wsb, B, H, W, K, N, st)
ap(); cv(); cx()
torch.cuda.synchronize()
print("outputs equal:", torch.equal(y1, y2))
t = [timeit(f) for f in (ap, cv, cx, wg, wx)]
print("32 -> 32 @ 128 x 128, batch 64: apply %.1f us | forward %.1f -> %.1f us (%+.0f %%) | filter gradient %.1f -> %.1f us (%+.0f %%) | apply + both %.1f -> %.1f us"
      % (t[0], t[1], t[2], 100 * (t[2] / t[1] - 1), t[3], t[4], 100 * (t[4] / t[3] - 1), t[0] + t[1] + t[3], t[2] + t[4]))
