"""TF-1.12 primitive semantics restated on torch-CPU (oracle; test infrastructure only).

Every function takes / returns NHWC torch tensors (fp64 or fp32) and is differentiable by
torch autograd, which is how the oracle obtains the gradients the reference gets from
``optimizer.minimize`` (phiseg/phiseg_model.py:141).  Reference call sites are cited per
function; the semantics marked [TF1.12] are the behaviour of the third-party TensorFlow 1.12
kernels behind those call sites.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import philox


# --------------------------------------------------------------------------------------------
# R1  tf.nn.conv2d(x, W, [1,1,1,1], 'SAME')   (tfwrapper/layers.py:123)  NHWC x HWIO, odd kernels
def conv2d_same(x, w_hwio):
    kh, kw = int(w_hwio.shape[0]), int(w_hwio.shape[1])
    assert kh % 2 == 1 and kw % 2 == 1, "only odd kernels occur on the hot path"
    y = F.conv2d(x.permute(0, 3, 1, 2), w_hwio.permute(3, 2, 0, 1), padding=(kh // 2, kw // 2))
    return y.permute(0, 2, 3, 1)


# tf.nn.bias_add (tfwrapper/layers.py:132)
def bias_add(x, b):
    return x + b.reshape(1, 1, 1, -1)


# --------------------------------------------------------------------------------------------
# R2  tf.contrib.layers.batch_norm(decay=.99, epsilon=1e-3, center, scale)  (tfwrapper/normalisation.py:156)
BN_EPS = 1e-3
BN_DECAY = 0.99


def batch_norm_train(x, gamma, beta):
    """[TF1.12] fused batch norm, training: biased batch variance normalises; returns
    (y, batch_mean, batch_var_unbiased) -- the unbiased variance feeds the moving average."""
    n = x.shape[0] * x.shape[1] * x.shape[2]
    mean = x.mean(dim=(0, 1, 2))
    var = ((x - mean) ** 2).mean(dim=(0, 1, 2))
    y = (x - mean) * torch.rsqrt(var + BN_EPS) * gamma + beta
    var_unbiased = var * (n / max(n - 1, 1))
    return y, mean, var_unbiased


def batch_norm_moving_update(moving, batch_value, decay=BN_DECAY):
    """[TF1.12] assign_moving_average without zero-debias: v <- v - (v - batch) * (1 - decay)."""
    return moving - (moving - batch_value) * (1.0 - decay)


def batch_norm_infer(x, gamma, beta, moving_mean, moving_var):
    return (x - moving_mean) * torch.rsqrt(moving_var + BN_EPS) * gamma + beta


# R3  group_norm2D (tfwrapper/normalisation.py:17-36)
def group_norm(x, gamma, beta, num_groups=None, eps=1e-5):
    n, h, w, c = x.shape
    g = num_groups if num_groups is not None else max(2, c // 16)
    xr = x.reshape(n, h, w, g, c // g)
    mean = xr.mean(dim=(1, 2, 4), keepdim=True)
    var = ((xr - mean) ** 2).mean(dim=(1, 2, 4), keepdim=True)
    xr = (xr - mean) / torch.sqrt(var + eps)
    return xr.reshape(n, h, w, c) * gamma.reshape(1, 1, 1, c) + beta.reshape(1, 1, 1, c)


# R3  instance_norm2D (tfwrapper/normalisation.py:3-14)
def instance_norm(x, scale, offset, eps=1e-5):
    mean = x.mean(dim=(1, 2), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(1, 2), keepdim=True)
    return scale * ((x - mean) * torch.rsqrt(var + eps)) + offset


# --------------------------------------------------------------------------------------------
# R4  tf.nn.avg_pool 2x2 stride 2 SAME (tfwrapper/layers.py:44-54)
def avg_pool_2x2_same(x):
    """[TF1.12] SAME pads bottom/right for odd sizes and divides by the number of VALID taps."""
    n, h, w, c = x.shape
    ph, pw = h % 2, w % 2
    xp = F.pad(x.permute(0, 3, 1, 2), (0, pw, 0, ph))
    s = F.avg_pool2d(xp, 2, 2) * 4.0
    ones = F.pad(torch.ones(1, 1, h, w, dtype=x.dtype), (0, pw, 0, ph))
    cnt = F.avg_pool2d(ones, 2, 2) * 4.0
    return (s / cnt).permute(0, 2, 3, 1)


# R5  tf.image.resize_images(x, [oh, ow])  = ResizeBilinear(align_corners=False), legacy coordinates
def _legacy_taps(in_size, out_size, dtype):
    scale = in_size / out_size
    src = torch.arange(out_size, dtype=torch.float64) * scale
    lo = torch.floor(src).to(torch.long)
    hi = torch.clamp(lo + 1, max=in_size - 1)
    frac = (src - lo.to(torch.float64)).to(dtype)
    return lo, hi, frac


def resize_bilinear_legacy(x, out_h, out_w):
    """[TF1.12] src = dst * (in/out) (NO half-pixel centres), lo=floor(src), hi=min(lo+1,in-1).
    Call site: tfwrapper/layers.py:336-345 (bilinear_upsample2D, factor 2)."""
    n, h, w, c = x.shape
    ylo, yhi, yf = _legacy_taps(h, out_h, x.dtype)
    xlo, xhi, xf = _legacy_taps(w, out_w, x.dtype)
    top = x[:, ylo]
    bot = x[:, yhi]

    def lerp_x(t):
        return t[:, :, xlo] + (t[:, :, xhi] - t[:, :, xlo]) * xf.reshape(1, 1, -1, 1)

    t, b = lerp_x(top), lerp_x(bot)
    return t + (b - t) * yf.reshape(1, -1, 1, 1)


# R6  tf.image.resize_images(..., NEAREST_NEIGHBOR)  (phiseg/model_zoo/likelihoods.py:221)
def resize_nearest(x, out_h, out_w):
    """[TF1.12] src = min(floor(dst * in/out), in-1)."""
    n, h, w, c = x.shape
    ys = torch.clamp(torch.floor(torch.arange(out_h, dtype=torch.float64) * (h / out_h)).to(torch.long), max=h - 1)
    xs = torch.clamp(torch.floor(torch.arange(out_w, dtype=torch.float64) * (w / out_w)).to(torch.long), max=w - 1)
    return x[:, ys][:, :, xs]


# --------------------------------------------------------------------------------------------
def softplus(x):          # tf.nn.softplus (posteriors.py:107,127; priors.py:99,119)
    return F.softplus(x)


def relu(x):              # tf.nn.relu (tfwrapper/layers.py:14)
    return torch.relu(x)


def one_hot(s, depth, dtype):   # tf.one_hot (phiseg_model.py:29)
    return F.one_hot(s.to(torch.long), depth).to(dtype)


def global_average_pool(x):     # tf.reduce_mean(x, axis=(1,2)) (tfwrapper/layers.py:70-78)
    return x.mean(dim=(1, 2))


# R10 multinoulli_loss_with_logits (phiseg_model.py:229-238): mean_b sum_pixels CE
def multinoulli_loss_with_logits(labels_oh, logits):
    bs = logits.shape[0]
    c = logits.shape[-1]
    lf = logits.reshape(bs, -1, c)
    yf = labels_oh.reshape(bs, -1, c)
    ce = -(yf * torch.log_softmax(lf, dim=-1)).sum(dim=-1)
    return ce.sum(dim=1).mean()


# R11 KL_two_gauss_with_diag_cov (phiseg_model.py:210-226)
def kl_two_gauss_with_diag_cov(mu0, sigma0, mu1, sigma1):
    bs = mu0.shape[0]
    s0 = sigma0.reshape(bs, -1) ** 2
    s1 = sigma1.reshape(bs, -1) ** 2
    m0 = mu0.reshape(bs, -1)
    m1 = mu1.reshape(bs, -1)
    t = (s0 + (m1 - m0) ** 2) / (s1 + 1e-10) + torch.log(s1 + 1e-10) - torch.log(s0 + 1e-10) - 1.0
    return (0.5 * t.sum(dim=1)).mean()


# R13 tf.train.AdamOptimizer (phiseg_model.py:137-141) -- [TF1.12] epsilon-hat form
def adam_tf1_step(p, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """t is the 1-based step.  lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps)."""
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    m = m + (g - m) * (1.0 - beta1)
    v = v + (g * g - v) * (1.0 - beta2)
    p = p - lr_t * m / (torch.sqrt(v) + eps)
    return p, m, v


# R14 he_normal = variance_scaling_initializer(factor=2, FAN_IN, uniform=False) (tfwrapper/utils.py:225-226)
def he_normal_truncated(shape, seed, stream):
    """[TF1.12] truncated normal (|n| <= 2 sigma by rejection), sigma = sqrt(1.3*2/fan_in),
    fan_in = kh*kw*Cin.  The draw order is the build's own (Philox stream), TF's RNG cannot
    be reproduced."""
    kh, kw, cin, cout = shape
    n = kh * kw * cin * cout
    std = math.sqrt(1.3 * 2.0 / (kh * kw * cin))
    raw = philox.normal(seed, 0, stream, 1, 2 * n + 64, dtype=np.float64)[0]
    keep = raw[np.abs(raw) <= 2.0]
    assert keep.size >= n
    return (keep[:n] * std).reshape(shape)


def conv2d_transpose_same(x, w_hwoi, stride):
    """[TF1.12] tf.nn.conv2d_transpose(x, filter [kh, kw, Cout, Cin], output [B, H*s, W*s, Cout], strides s, 'SAME')
    (tfwrapper/layers.py:225-229): the gradient of the stride-s SAME conv2d with respect to its input -- torch's
    conv_transpose2d gives the un-cropped result of size (H - 1) s + k, of which rows / columns [pad_before, pad_before + H s)
    are kept, pad_before = max((H - 1) s + k - H s, 0) // 2 (TF's SAME rule for the forward convolution)."""
    kh, kw, cout, cin = w_hwoi.shape
    sh, sw = stride
    B, H, W, _ = x.shape
    full = F.conv_transpose2d(x.permute(0, 3, 1, 2), w_hwoi.permute(3, 2, 0, 1), stride=(sh, sw))
    Ho, Wo = H * sh, W * sw
    th, tw = max((H - 1) * sh + kh - Ho, 0), max((W - 1) * sw + kw - Wo, 0)
    pt, pl = th // 2, tw // 2
    full = F.pad(full, (0, max(0, pl + Wo - full.shape[3]), 0, max(0, pt + Ho - full.shape[2])))
    return full[:, :, pt:pt + Ho, pl:pl + Wo].permute(0, 2, 3, 1)


# --------------------------------------------------------------------------------------------
# (f)4  layers without a call site in the shipped experiments
def conv2d_general_same(x, w_hwio, stride=(1, 1), dilation=(1, 1)):
    """[TF1.12] tf.nn.conv2d(x, W, [1, sh, sw, 1], 'SAME') (conv2D with strides, tfwrapper/layers.py:123) and
    tf.nn.atrous_conv2d(x, W, rate, 'SAME') (dilated_conv2D, layers.py:404): output ceil(H / s); with the effective kernel extent
    ke = (k - 1) d + 1 the total padding is max((Ho - 1) s + ke - H, 0), the smaller half in front."""
    kh, kw = int(w_hwio.shape[0]), int(w_hwio.shape[1])
    (sh, sw), (dh, dw) = stride, dilation
    B, H, W, _ = x.shape
    Ho, Wo = -(-H // sh), -(-W // sw)
    th = max((Ho - 1) * sh + (kh - 1) * dh + 1 - H, 0)
    tw = max((Wo - 1) * sw + (kw - 1) * dw + 1 - W, 0)
    xp = F.pad(x.permute(0, 3, 1, 2), (tw // 2, tw - tw // 2, th // 2, th - th // 2))
    y = F.conv2d(xp, w_hwio.permute(3, 2, 0, 1), stride=(sh, sw), dilation=(dh, dw))
    return y.permute(0, 2, 3, 1)


def max_pool_2x2_same(x):
    """[TF1.12] tf.nn.max_pool(x, [1,2,2,1], [1,2,2,1], 'SAME') (maxpool2D, layers.py:18-28): odd sizes are padded at the bottom /
    right and the padding never wins (-inf); the gradient goes to the first maximum of a window (row-major), as in TF's and
    torch's kernels."""
    n, h, w, c = x.shape
    xp = F.pad(x.permute(0, 3, 1, 2), (0, w % 2, 0, h % 2), value=float("-inf"))
    return F.max_pool2d(xp, 2, 2).permute(0, 2, 3, 1)


def spatial_window(x, out_h, out_w, off_y, off_x):
    """dst[b, y, x] = src[b, y + off_y, x + off_x] or 0: pad_to_size (layers.py:625-650: off = -(size_diff // 2)) and the centre
    crop of crop_and_concat (layers.py:586-622: off = (larger - output) // 2)."""
    B, H, W, C = x.shape
    out = torch.zeros(B, out_h, out_w, C, dtype=x.dtype)
    ys = [y for y in range(out_h) if 0 <= y + off_y < H]
    xs = [xx for xx in range(out_w) if 0 <= xx + off_x < W]
    if ys and xs:
        out[:, ys[0]:ys[-1] + 1, xs[0]:xs[-1] + 1] = x[:, ys[0] + off_y:ys[-1] + off_y + 1, xs[0] + off_x:xs[-1] + off_x + 1]
    return out


def dropout_keep_mask(shape, keep_prob, seed, step, stream, sample_offset=0):
    """Keep mask of the build's dropout contract (tf.nn.dropout itself is unseeded in the reference): element e of
    sample b keeps iff the 24-bit uniform (word e % 4 of Philox block e // 4 under (seed, step, stream, sample_offset + b)) >> 8
    / 2^24 is below keep_prob."""
    import numpy as np
    from oracle import philox
    B, per = int(shape[0]), int(np.prod(shape[1:]))
    nblk = (per + 3) // 4
    out = np.zeros((B, nblk * 4), dtype=bool)
    for b in range(B):
        ctr = np.zeros((nblk, 4), dtype=np.uint32)
        ctr[:, 0] = np.arange(nblk)
        ctr[:, 1] = sample_offset + b
        ctr[:, 2] = stream
        ctr[:, 3] = step
        words = philox.philox4x32_10(ctr, np.array([seed & 0xffffffff, (seed >> 32) & 0xffffffff], dtype=np.uint32))
        u = (words >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
        out[b] = (u < np.float32(keep_prob)).reshape(-1)
    return out[:, :per].reshape(shape)
