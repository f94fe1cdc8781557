"""(f)4 transposed_conv2D (reference tfwrapper/layers.py:197-258, tf.nn.conv2d_transpose) -- oracle pinned by the DEFINITION
of the op (the gradient of the stride-s SAME conv2d with respect to its input), HIP kernels and the graph-level layer against it."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tf1_ops as T

CASES = [(2, 5, 6, 3, 4, 4, 4, 2, 2), (1, 4, 4, 8, 2, 3, 3, 2, 2), (2, 3, 5, 2, 3, 2, 2, 2, 2), (1, 4, 3, 2, 2, 5, 5, 3, 3),
         (2, 8, 8, 16, 8, 4, 4, 2, 2)]      # B, H, W, Cin, Cout, kh, kw, sh, sw


def _same_conv_strided(z, w_hwoi, stride):
    """The forward convolution conv2d_transpose is the input-gradient of: z [B,Ho,Wo,Cout] -> [B,H,W,Cin], filter read as
    [kh, kw, in = Cout, out = Cin], strides s, SAME (TF pads pad_total // 2 before, the rest after)."""
    kh, kw, cout, cin = w_hwoi.shape
    sh, sw = stride
    Ho, Wo = z.shape[1], z.shape[2]
    H, W = -(-Ho // sh), -(-Wo // sw)
    th, tw = max((H - 1) * sh + kh - Ho, 0), max((W - 1) * sw + kw - Wo, 0)
    zp = F.pad(z.permute(0, 3, 1, 2), (tw // 2, tw - tw // 2, th // 2, th - th // 2))
    return F.conv2d(zp, w_hwoi.permute(3, 2, 0, 1), stride=(sh, sw)).permute(0, 2, 3, 1)


@pytest.mark.parametrize("case", CASES)
def test_oracle_conv2d_transpose_is_the_input_gradient_of_conv2d(case):
    B, H, W, Cin, Cout, kh, kw, sh, sw = case
    rng = np.random.default_rng(0)
    x = torch.as_tensor(rng.standard_normal((B, H, W, Cin)))
    w = torch.as_tensor(rng.standard_normal((kh, kw, Cout, Cin)))
    z = torch.zeros(B, H * sh, W * sw, Cout, dtype=torch.float64, requires_grad=True)
    y = _same_conv_strided(z, w, (sh, sw))
    assert y.shape == x.shape
    (y * x).sum().backward()
    np.testing.assert_allclose(T.conv2d_transpose_same(x, w, (sh, sw)).numpy(), z.grad.numpy(), rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_tconv_kernels_match_oracle(case, dt):
    from tests.test_kernels_gpu import BF16, F32, S, close, dev, host, rounded, tdt
    from phiseg_code_amd import runtime as rt
    L = rt.lib()
    code = F32 if dt == "f32" else BF16
    B, H, W, Cin, Cout, kh, kw, sh, sw = case
    rng = np.random.default_rng(1)
    x = rng.standard_normal((B, H, W, Cin))
    w = rng.standard_normal((kh, kw, Cout, Cin)) / np.sqrt(kh * kw * Cin)
    b = rng.standard_normal(Cout) * 0.3
    xr = rounded(x, code).requires_grad_(True)
    wr = torch.as_tensor(w, dtype=torch.float32).double().requires_grad_(True)
    br = torch.as_tensor(b, dtype=torch.float32).double().requires_grad_(True)
    pre = T.conv2d_transpose_same(xr, wr, (sh, sw)) + br
    yr = T.relu(pre)
    xd, wd, bd = dev(x, code), dev(w), dev(b)
    y = torch.empty(B, H * sh, W * sw, Cout, dtype=tdt(code)).cuda()
    L.tconv2d_fwd(xd.data_ptr(), code, wd.data_ptr(), bd.data_ptr(), y.data_ptr(), code, B, H, W, Cin, Cout, kh, kw, sh, sw, 1, S())
    tol = 1e-5 if dt == "f32" else 8e-3
    close(host(y), yr.detach().numpy(), tol, "tconv fwd")
    dy = rng.standard_normal((B, H * sh, W * sw, Cout))
    dyr = rounded(dy, code)
    (pre * dyr).sum().backward()
    dyd = dev(dy, code)
    dx = torch.empty(B, H, W, Cin, dtype=tdt(code)).cuda()
    L.tconv2d_dgrad(dyd.data_ptr(), code, wd.data_ptr(), dx.data_ptr(), code, B, H, W, Cin, Cout, kh, kw, sh, sw, S())
    close(host(dx), xr.grad.numpy(), tol, "tconv dgrad")
    dw = torch.zeros(kh, kw, Cout, Cin, dtype=torch.float32).cuda()
    L.tconv2d_wgrad(xd.data_ptr(), code, dyd.data_ptr(), code, dw.data_ptr(), B, H, W, Cin, Cout, kh, kw, sh, sw, S())
    close(host(dw), wr.grad.numpy(), 2e-5, "tconv wgrad")


@pytest.mark.gpu
@pytest.mark.parametrize("norm", ["identity", "batch_norm", "group_norm2D"])
def test_transposed_conv2D_layer_in_a_graph(norm):
    """The layer as the reference's code would call it -- conv2D -> transposed_conv2D(4x4, stride 2, norm, relu) -> 1x1 head ->
    cross-entropy at the up-sampled resolution -- compiled by the engine: loss and every gradient vs torch autograd of the
    oracle primitives."""
    from phiseg_code_amd import engine
    from phiseg_code_amd import graph as G
    from phiseg_code_amd.tfwrapper import activations as act
    from phiseg_code_amd.tfwrapper import layers
    from phiseg_code_amd.tfwrapper import normalisation as tfnorm
    B, H, C = 2, 8, 2
    g = G.reset_default_graph()
    x_inp = G.placeholder(G.KIND_F32, [None, H, H, 3], name="x_input")
    s_inp = G.placeholder(G.KIND_U8, [None, 2 * H, 2 * H], name="s_input")
    nfn = getattr(tfnorm, norm)
    with g.variable_scope("net"):
        h = layers.conv2D(x_inp, "c1", num_filters=8, normalisation=nfn, training=True)
        u = layers.transposed_conv2D(h, "up", num_filters=6, normalisation=nfn, training=True)
        s = layers.conv2D(u, "head", num_filters=C, kernel_size=(1, 1), activation=act.identity)
    ce, _ = G.residual_multinoulli([s], s_inp, 1.0)
    loss = G.weighted_sum([ce[0]], [1.0])
    store = engine.ParamStore(g, seed=3)
    rng = np.random.default_rng(7)
    vals = {n: (v.initial_value(3) + (0.1 * rng.standard_normal(v.shape) if not n.endswith("/W") else 0)).astype(np.float32)
            for n, v in g.variables.items()}
    for n in vals:
        if n.endswith("moving_variance"):
            vals[n] = np.abs(vals[n]) + 0.5
    store.load(vals)
    plan = engine.Plan(store, [loss, s], loss=loss, batch=B, training=True, compute_dtype="f32", optimize=False, use_hip_graph=False)
    x = rng.standard_normal((B, H, H, 3)).astype(np.float32)
    lab = rng.integers(0, C, (B, 2 * H, 2 * H)).astype(np.uint8)
    plan.set_input("x_input", x)
    plan.set_input("s_input", lab)
    plan.run(sync=True)
    got_loss, got_s = float(plan.fetch(loss)), plan.fetch(s)
    grads = store.export(grads=True)
    # oracle
    p = {n: torch.as_tensor(v, dtype=torch.float64).requires_grad_(not n.rsplit("/", 1)[-1].startswith("moving_")) for n, v in vals.items()}

    def nrm(t, scope, kind):
        if kind == "identity":
            return t
        if kind == "batch_norm":
            y, _, _ = T.batch_norm_train(t, p[scope + "/batch_norm/BatchNorm/gamma"], p[scope + "/batch_norm/BatchNorm/beta"])
            return y
        return T.group_norm(t, p[scope + "/group_norm/gamma"], p[scope + "/group_norm/beta"], max(2, t.shape[-1] // 16))
    xt = torch.as_tensor(x, dtype=torch.float64)
    h = T.conv2d_same(xt, p["net/c1/W"])
    if norm != "batch_norm":
        h = T.bias_add(h, p["net/c1/b"])
    h = T.relu(nrm(h, "net/c1", norm))
    u = T.conv2d_transpose_same(h, p["net/up/W"], (2, 2)) + p["net/up/b"]          # the bias stays even in front of batch norm
    u = T.relu(nrm(u, "net/up", norm))
    so = T.bias_add(T.conv2d_same(u, p["net/head/W"]), p["net/head/b"])
    ref = T.multinoulli_loss_with_logits(T.one_hot(torch.as_tensor(lab), C, torch.float64), so)
    ref.backward()
    np.testing.assert_allclose(got_s, so.detach().numpy(), rtol=0, atol=2e-4 * float(so.abs().max()))
    np.testing.assert_allclose(got_loss, float(ref), rtol=2e-5)
    for n, t in p.items():
        if t.grad is None:
            continue
        r = t.grad.numpy()
        if np.abs(r).max() < 1e-9:          # a bias in front of batch norm: its gradient is exactly zero, fp32 leaves round-off
            assert np.abs(grads[n]).max() < 1e-4, n
            continue
        np.testing.assert_allclose(grads[n], r, rtol=0, atol=3e-3 * max(np.abs(r).max(), 1e-6), err_msg=n)
