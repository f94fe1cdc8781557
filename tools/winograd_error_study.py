"""GPU box (or CPU): what would a Winograd F(2x2, 3x3) path cost in accuracy with bf16 MFMA operands?
Direct path (what the kernels do): bf16 x, bf16 w, exact products, fp32 accumulation, bf16 output.
Winograd path: V = B^T d B and U = G g G^T computed in fp32 from the bf16 tensors and ROUNDED to bf16 (the MFMA operands), the 16
element-wise GEMMs accumulated in fp32, Y = A^T M A in fp32, bf16 output.  Errors against the float64 convolution of the same bf16
tensors, relative to the largest output (the kernel tests' measure) and as RMS relative to the output's RMS.
usage: python tools/winograd_error_study.py [K N H B]"""
import sys
import torch

dev = "cuda" if torch.cuda.is_available() else "cpu"
K, N, H, B = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (128, 128, 32, 4)
g = torch.Generator(device=dev).manual_seed(3)
bf = lambda t: t.to(torch.bfloat16).to(torch.float64)
x = bf(torch.relu(torch.randn(B, K, H, H, device=dev, generator=g)))                    # post-ReLU activations
w = bf(torch.randn(N, K, 3, 3, device=dev, generator=g) / (9 * K) ** 0.5)
ref = torch.nn.functional.conv2d(x, w, padding=1)
direct = bf(ref)                                                                         # fp32 accumulation error is negligible beside this

Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64, device=dev)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64, device=dev)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64, device=dev)
xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                               # [B, K, H/2, H/2, 4, 4]
V = torch.einsum("ij,bkyxjl,ml->bkyxim", Bt, tiles, Bt)
U = torch.einsum("ij,nkjl,ml->nkim", G, w, G)
res = {}
for name, rv, ru in (("fp32 transformed operands (no rounding)", False, False), ("bf16 V, fp32 U", True, False), ("bf16 V and U (MFMA operands)", True, True)):
    Vq, Uq = (bf(V) if rv else V), (bf(U) if ru else U)
    M = torch.einsum("bkyxim,nkim->bnyxim", Vq, Uq)
    Y = torch.einsum("ij,bnyxjl,ml->bnyxim", At, M, At)                                   # [B, N, H/2, H/2, 2, 2]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(B, N, H, H)
    res[name] = bf(y)
mx, rms = float(ref.abs().max()), float((ref ** 2).mean().sqrt())
print("K = %d, N = %d, %d x %d, batch %d; output max %.3f, rms %.3f" % (K, N, H, H, B, mx, rms))
e = direct - ref
print("%-46s max err / max %.2e   rms err / rms %.2e" % ("direct, bf16 output", float(e.abs().max()) / mx, float((e ** 2).mean().sqrt()) / rms))
for name, y in res.items():
    e = y - ref
    print("%-46s max err / max %.2e   rms err / rms %.2e" % ("Winograd, " + name, float(e.abs().max()) / mx, float((e ** 2).mean().sqrt()) / rms))
