"""GPU box, round-5 spike: the pair kernel with scale-shift-ReLU in its LOADER half (phx_conv3x3_mfma_bf16_xf) against the LDS-DMA
loader on the materialised a = relu(x * scale + shift), and the apply pass it would delete.  Each launch alone (HIP events).
usage: python tools/bench_pp_xf.py"""
import sys
import torch
sys.path.insert(0, ".")
from phiseg_code_amd import runtime as rt
L = rt.debug_lib()
st = torch.cuda.current_stream().cuda_stream
BF = rt.BF16


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (B, H, W, K, N) in [(64, 128, 128, 128, 128), (64, 64, 64, 192, 192), (64, 128, 128, 64, 128), (64, 64, 64, 64, 64), (64, 64, 64, 128, 192)]:
    x = torch.randn(B, H, W, K, device="cuda").to(torch.bfloat16)
    sc = (1.0 + 0.2 * torch.randn(K, device="cuda")).float().contiguous()
    sh = (0.1 * torch.randn(K, device="cuda")).float().contiguous()
    a = torch.empty_like(x)
    wf = (torch.randn(9 * K * N, device="cuda") * 0.05).to(torch.bfloat16)
    y1 = torch.empty(B, H, W, N, device="cuda", dtype=torch.bfloat16)
    y2 = torch.empty_like(y1)
    P = B * H * W
    L.debug_conv_policy(2, 1)
    nt = L.conv3x3_mfma_bf16_tiles(B, H, W, K, N)
    part1 = torch.zeros(nt * 2 * N, device="cuda")
    part2 = torch.zeros(nt * 2 * N, device="cuda")
    apply_ = lambda: L.affine_act(x.data_ptr(), BF, sc.data_ptr(), sh.data_ptr(), a.data_ptr(), BF, 1, P, K, 1, st)
    dma = lambda: L.conv3x3_mfma_bf16(a.data_ptr(), wf.data_ptr(), y1.data_ptr(), None, 0, part1.data_ptr(), B, H, W, K, N, st)
    xf = lambda: L.conv3x3_mfma_bf16_xf(x.data_ptr(), sc.data_ptr(), sh.data_ptr(), wf.data_ptr(), y2.data_ptr(), part2.data_ptr(), B, H, W, K, N, st)
    apply_(); dma(); xf()
    torch.cuda.synchronize()
    same = torch.equal(y1, y2)
    err = (y1.float() - y2.float()).abs().max().item()
    t_ap, t_dma, t_xf = timeit(apply_), timeit(dma), timeit(xf)
    fl = 18.0 * K * N * P
    print("%-26s apply %.3f ms | conv (LDS-DMA loader) %.3f ms %4.0f TF | conv (transform in the loader) %.3f ms %4.0f TF (%+.0f %%) | "
          "apply + conv %.3f vs fused %.3f ms | outputs equal: %s (max abs diff %.3g)"
          % ((B, H, W, K, N), t_ap, t_dma, fl / t_dma / 1e9, t_xf, fl / t_xf / 1e9, 100 * (t_xf / t_dma - 1), t_ap + t_dma, t_xf, same, err), flush=True)
    if K % 64 == 0 and N % 64 == 0 and L.conv3x3_wgrad_xf_supported(B, H, W, K, N):
        dy = torch.randn(B, H, W, N, device="cuda").to(torch.bfloat16)
        wsb = int(L.conv3x3_wgrad_ws_bytes(B, H, W, K, N))
        ws = torch.empty(wsb // 4, device="cuda")
        dw = torch.zeros(9 * K * N, device="cuda")
        wg = lambda: L.conv3x3_wgrad_mfma_bf16_partial(a.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), wsb, B, H, W, K, N, st)
        wgx = lambda: L.conv3x3_wgrad_mfma_bf16_partial_xf(x.data_ptr(), sc.data_ptr(), sh.data_ptr(), dy.data_ptr(), ws.data_ptr(), wsb, B, H, W, K, N, st)
        t_w, t_wx = timeit(wg), timeit(wgx)
        print("%-26s filter gradient (stand-alone launch, no reduction): LDS-DMA loader %.3f ms | transform in the loader %.3f ms (%+.0f %%)"
              % ("", t_w, t_wx, 100 * (t_wx / t_w - 1)), flush=True)
L.debug_conv_policy(1, 1)
