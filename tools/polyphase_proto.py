"""CPU prototype (float64, torch) of the identity round 6 starts from ("the next lever"):

    conv2d_3x3_SAME(resize_bilinear_legacy_x2(x), W)  ==  depth_to_space_2( conv2d_3x3_SAME(x, Weff) )  +  border terms

for layers.bilinear_upsample2D -> layers.conv2D (likelihoods.py:166-170, 200-204; TF 1.12 legacy resize: out[2k] = in[k],
out[2k+1] = (in[k] + in[min(k+1, n-1)]) / 2 per axis).  Weff[dh, dw, ci, (a, b, co)] = sum_{kh,kw} My[a, dh, kh] Mx[b, dw, kw] W[kh, kw, ci, co]
with the per-axis phase tables M below: a 3x3 convolution at the LOW resolution with 4 x Cout output channels.  The borders differ
because conv2D zero-pads the UP-SAMPLED map and the resize clamps at the far edge: per axis, with zero-padded x,

    hi position 0       : the formula adds  (w[-1] / 2) x[0]        that is not there  (up-sampled position -1 is padding, not (x[-1] + x[0]) / 2)
    hi position 2n - 2  : the formula lacks (w[+1] / 2) x[n-1]      (up-sampled position 2n-1 is x[n-1], not (x[n-1] + 0) / 2)
    hi position 2n - 1  : the formula lacks (w[ 0] / 2) x[n-1]      (same position, centre tap)

i.e. the exact 1-D operator is A_k = Atilde_k + D_k with D_k non-zero in three rows, and in 2-D
    Y = sum_{kh,kw} W[kh,kw] (Atilde + D)_kh (x) (Atilde + D)_kw x
      = main + sum W D_kh (x) Atilde_kw x + sum W Atilde_kh (x) D_kw x + sum W D_kh (x) D_kw x      (rows, columns, corners).
This script builds the dense 1-D operators from the oracle's resize, extracts Atilde / D, checks the tables M and the three
correction coefficients against them, and checks the 2-D identity (forward value, and the gradients with respect to x and W through
autograd of the decomposed form) against the oracle's composition.  usage: python tools/polyphase_proto.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import tf1_ops as T  # noqa: E402

torch.manual_seed(0)
D64 = torch.float64
# phase tables: M[a][dh][kh], dh / kh in {-1, 0, +1} -> index + 1
M = torch.zeros(2, 3, 3, dtype=D64)
M[0] = torch.tensor([[0.5, 0.0, 0.0], [0.5, 1.0, 0.5], [0.0, 0.0, 0.5]])       # even output: x[i-1], x[i], x[i+1]
M[1] = torch.tensor([[0.0, 0.0, 0.0], [1.0, 0.5, 0.0], [0.0, 0.5, 1.0]])       # odd output:        x[i], x[i+1]


def dense_1d_ops(n):
    """A[k] (2n x n): up-sampled, zero-padded signal shifted by tap k in {-1, 0, 1}: (A[k] x)[q] = u[q + k], u = resize(x), u[-1] = u[2n] = 0."""
    eye = torch.eye(n, dtype=D64).reshape(n, n, 1, 1)                          # n "images" of height n, width 1
    u = T.resize_bilinear_legacy(eye, 2 * n, 1).reshape(n, 2 * n).t()          # (2n x n): u = U x
    up = torch.zeros(2 * n + 2, n, dtype=D64)
    up[1:-1] = u
    return [up[1 + k: 1 + k + 2 * n] for k in (-1, 0, 1)]


def formula_1d_ops(n):
    """Atilde[k]: the phase formula on zero-padded x: (Atilde[k] x)[2i + a] = sum_dh M[a][dh][k] x[i + dh]."""
    out = []
    for k in range(3):
        A = torch.zeros(2 * n, n, dtype=D64)
        for i in range(n):
            for a in range(2):
                for dh in (-1, 0, 1):
                    if 0 <= i + dh < n:
                        A[2 * i + a, i + dh] += M[a, dh + 1, k]
        out.append(A)
    return out


def check_1d(n):
    A, At = dense_1d_ops(n), formula_1d_ops(n)
    for k in range(3):
        Dk = A[k] - At[k]
        exp = torch.zeros_like(Dk)
        if k == 0:
            exp[0, 0] = -0.5                   # tap -1 at hi position 0
        if k == 2:
            exp[2 * n - 2, n - 1] = 0.5        # tap +1 at hi position 2n - 2
        if k == 1:
            exp[2 * n - 1, n - 1] = 0.5        # tap 0 at hi position 2n - 1
        assert torch.allclose(Dk, exp, atol=1e-14), (n, k, (Dk - exp).abs().max())
    return A, At


def weff(W):
    """[3, 3, Ci, Co] -> [3, 3, Ci, 4 Co] (output channel (a * 2 + b) * Co + co)"""
    Ci, Co = W.shape[2], W.shape[3]
    We = torch.einsum("ahk,bwl,klio->hwiabo", M, M, W)                          # dh, dw, ci, a, b, co
    return We.reshape(3, 3, Ci, 4 * Co)


def depth_to_space(y, Co):
    B, h, w, _ = y.shape
    return y.reshape(B, h, w, 2, 2, Co).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * h, 2 * w, Co)


def border_terms(x, W, A_h, At_h, A_w, At_w):
    """rows + columns + corners from the dense 1-D operators (the three non-zero rows of D per axis); W[kh, kw, ci, co]"""
    Dh = [a - b for a, b in zip(A_h, At_h)]
    Dw = [a - b for a, b in zip(A_w, At_w)]
    y = 0.0
    for kh in range(3):
        for kw in range(3):
            for Lh, Lw in ((Dh[kh], At_w[kw]), (At_h[kh], Dw[kw]), (Dh[kh], Dw[kw])):
                t = torch.einsum("qh,bhwi->bqwi", Lh, x)
                t = torch.einsum("pw,bqwi->bqpi", Lw, t)
                y = y + torch.einsum("bqpi,io->bqpo", t, W[kh, kw])
    return y


def frame_form(x, W):
    """The form a first implementation can take with NO new matrix kernel: the phase convolution gives every
    hi-res pixel except the frame (rows / columns 0, 2n - 2, 2n - 1) exactly; the frame comes from the ordinary 3x3 convolution run on
    two small gathered images -- per image six rows (one zero row, u[0], u[1], u[2n - 3], u[2n - 2], u[2n - 1]) stacked into ONE tall
    image [1, 6 B, 2w, Ci] (the zero row is the padding between neighbours; outputs of rows 1, 4, 5 of each group of six are hi rows
    0, 2n - 2, 2n - 1), and the same for the columns with the filter transposed.  Returns the assembled hi-res map."""
    B, h, w, Ci = x.shape
    Co = W.shape[3]
    u = T.resize_bilinear_legacy(x, 2 * h, 2 * w)                     # (only its six frame rows / columns are gathered: 6 / 2n of it)
    y = depth_to_space(T.conv2d_same(x, weff(W)), Co)
    zr = torch.zeros(B, 1, 2 * w, Ci, dtype=D64)
    rows = torch.cat([zr, u[:, 0:2], u[:, 2 * h - 3:2 * h]], dim=1).reshape(1, 6 * B, 2 * w, Ci)
    fr = T.conv2d_same(rows, W).reshape(B, 6, 2 * w, Co)
    ut = u.transpose(1, 2)                                             # columns as rows
    zc = torch.zeros(B, 1, 2 * h, Ci, dtype=D64)
    cols = torch.cat([zc, ut[:, 0:2], ut[:, 2 * w - 3:2 * w]], dim=1).reshape(1, 6 * B, 2 * h, Ci)
    fc = T.conv2d_same(cols, W.transpose(0, 1)).reshape(B, 6, 2 * h, Co)
    out = y.clone()
    for src, dst in ((1, 0), (4, 2 * w - 2), (5, 2 * w - 1)):
        out[:, :, dst] = fc[:, src]                                   # frame columns (all rows; the corners are overwritten next)
    for src, dst in ((1, 0), (4, 2 * h - 2), (5, 2 * h - 1)):
        out[:, dst] = fr[:, src]                                      # frame rows
    return out, y


def main():
    for (B, h, w, Ci, Co) in [(2, 4, 6, 5, 3), (3, 8, 8, 4, 2)]:
        x = torch.randn(B, h, w, Ci, dtype=D64, requires_grad=True)
        W = torch.randn(3, 3, Ci, Co, dtype=D64, requires_grad=True)
        ref = T.conv2d_same(T.resize_bilinear_legacy(x, 2 * h, 2 * w), W)
        got, y_main = frame_form(x, W)
        inner = (y_main - ref)[:, 1:2 * h - 2, 1:2 * w - 2].abs().max().item()
        dy = torch.randn_like(ref)
        gx_r, gw_r = torch.autograd.grad((ref * dy).sum(), (x, W), retain_graph=True)
        gx_g, gw_g = torch.autograd.grad((got * dy).sum(), (x, W))
        print("frame form (B, h, w, Ci, Co) = %-16s interior of the phase convolution %.2e   assembled map %.2e   d/dx %.2e   d/dW %.2e"
              % ((B, h, w, Ci, Co), inner, (got - ref).abs().max().item(), (gx_g - gx_r).abs().max().item(), (gw_g - gw_r).abs().max().item()))
        assert inner < 1e-12 and (got - ref).abs().max() < 1e-12 and (gx_g - gx_r).abs().max() < 1e-11 and (gw_g - gw_r).abs().max() < 1e-11
    for n in (2, 3, 5, 8):
        check_1d(n)
    print("1-D: phase tables and the three border coefficients (-1/2 w[-1] x[0] at 0, +1/2 w[+1] x[n-1] at 2n-2, +1/2 w[0] x[n-1] at 2n-1) exact")
    for (B, h, w, Ci, Co) in [(2, 4, 6, 5, 3), (1, 8, 8, 4, 2), (3, 2, 3, 3, 4)]:
        x = torch.randn(B, h, w, Ci, dtype=D64, requires_grad=True)
        W = torch.randn(3, 3, Ci, Co, dtype=D64, requires_grad=True)
        ref = T.conv2d_same(T.resize_bilinear_legacy(x, 2 * h, 2 * w), W)
        A_h, At_h = check_1d(h)
        A_w, At_w = check_1d(w)
        main_term = depth_to_space(T.conv2d_same(x, weff(W)), Co)
        got = main_term + border_terms(x, W, A_h, At_h, A_w, At_w)
        err = (got - ref).abs().max().item()
        frac_border = ((main_term - ref).abs() > 1e-12).any(dim=-1).double().mean().item()
        dy = torch.randn_like(ref)
        gx_r, gw_r = torch.autograd.grad((ref * dy).sum(), (x, W), retain_graph=True)
        gx_g, gw_g = torch.autograd.grad((got * dy).sum(), (x, W))
        print("(B, h, w, Ci, Co) = %-18s forward max err %.2e   d/dx %.2e   d/dW %.2e   pixels the main term alone gets wrong: %.1f %% (the first row / column and the last two)"
              % ((B, h, w, Ci, Co), err, (gx_g - gx_r).abs().max().item(), (gw_g - gw_r).abs().max().item(), 100 * frac_border))
        assert err < 1e-12 and (gx_g - gx_r).abs().max() < 1e-11 and (gw_g - gw_r).abs().max() < 1e-11
    print("2-D identity (main + rows + columns + corners) exact in float64, values and both gradients")


if __name__ == "__main__":
    main()
