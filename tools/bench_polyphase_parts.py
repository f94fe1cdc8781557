"""GPU box: the matrix launches of the two forms of bilinear_upsample2D -> conv2D (192 -> 32, 64 x 64 -> 128 x 128, batch 64), each
alone (HIP events): today's (resize, 192 -> 32 @ 128 x 128 forward / data gradient / filter gradient, resize adjoint) against the phase
form's (192 -> 128 @ 64 x 64 forward / data gradient / filter gradient + the frame's 192 -> 32 @ [1, 384, 128])."""
import sys
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.lib()
st = torch.cuda.current_stream().cuda_stream
BF = rt.BF16


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def bf(*shape):
    return torch.randn(*shape, device="cuda").to(torch.bfloat16)


def conv_set(B, H, W, K, N, tag):
    x, dy = bf(B, H, W, K), bf(B, H, W, N)
    y, dx = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16), torch.empty(B, H, W, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(3, 3, K, N, device="cuda") * 0.05
    wf, wd = torch.empty(9 * K * N, device="cuda", dtype=torch.bfloat16), torch.empty(9 * K * N, device="cuda", dtype=torch.bfloat16)
    L.pack_conv3x3_bf16(w.data_ptr(), wf.data_ptr(), wd.data_ptr(), K, N, st)
    nt = L.conv3x3_mfma_bf16_tiles(B, H, W, K, N)
    part = torch.zeros(nt * 2 * N, device="cuda")
    wsb = int(L.conv3x3_wgrad_ws_bytes(B, H, W, K, N))
    ws = torch.empty(max(wsb, 4) // 4, device="cuda")
    dw = torch.zeros(9 * K * N, device="cuda")
    f = lambda: L.conv3x3_mfma_bf16(x.data_ptr(), wf.data_ptr(), y.data_ptr(), None, 0, part.data_ptr(), B, H, W, K, N, st)
    g = lambda: L.conv3x3_mfma_bf16(dy.data_ptr(), wd.data_ptr(), dx.data_ptr(), None, 0, None, B, H, W, N, K, st)
    h = lambda: L.conv3x3_wgrad_mfma_bf16(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), wsb, B, H, W, K, N, st)
    t = [timeit(f), timeit(g), timeit(h)]
    print("%-28s %3d -> %3d @ [%d, %d, %d]: forward %6.1f  data gradient %6.1f  filter gradient %6.1f us  (sum %.1f)" % (tag, K, N, B, H, W, t[0], t[1], t[2], sum(t)))
    return sum(t)


B = 64
x = bf(B, 64, 64, 192)
u = torch.empty(B, 128, 128, 192, device="cuda", dtype=torch.bfloat16)
tf = timeit(lambda: L.bilinear_up2x_fwd(x.data_ptr(), BF, u.data_ptr(), B, 64, 64, 192, st))
tb = timeit(lambda: L.bilinear_up2x_bwd(u.data_ptr(), BF, x.data_ptr(), B, 64, 64, 192, st))
print("resize 192 ch 64 -> 128: forward %.1f us, adjoint %.1f us" % (tf, tb))
now = conv_set(B, 128, 128, 192, 32, "today") + tf + tb
main = conv_set(B, 64, 64, 192, 128, "phase form, main")
fr = conv_set(1, 6 * B, 128, 192, 32, "phase form, frame (x 2)")
print("today %.1f us   phase form %.1f us + elementwise passes (gather / scatter / permutes / fold: ~0.6 GB ~ 120 us)" % (now, main + 2 * fr))
