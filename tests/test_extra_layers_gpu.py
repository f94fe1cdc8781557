"""layers of tfwrapper/layers.py without a call site in the shipped experiments (csrc/gconv.hip) --
strided / dilated convolution, max pool, pad_to_size / crop, dropout -- kernels through the C ABI against the oracle (torch
autograd of oracle/tf1_ops.py), fp32 and bf16 storage, odd sizes, then the layer functions through the engine."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import tf1_ops as T

pytestmark = pytest.mark.gpu
F32, BF16 = 0, 1
RNG = np.random.default_rng(11)


@pytest.fixture(scope="module")
def L():
    from phiseg_code_amd import runtime as rt
    return rt.lib()


def S():
    return torch.cuda.current_stream().cuda_stream


def tdt(dt):
    return torch.float32 if dt == F32 else torch.bfloat16


def dev(a, dt=F32):
    return torch.as_tensor(np.asarray(a, dtype=np.float32)).to(tdt(dt)).cuda()


def rounded(a, dt):
    return torch.as_tensor(np.asarray(a, dtype=np.float32)).to(tdt(dt)).double()


def close(a, b, tol, what=""):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b).max() / max(1e-12, np.abs(b).max())
    assert err < tol, (what, err)


@pytest.mark.parametrize("case", [(2, 9, 7, 5, 6, 3, 3, 1, 1, 2, 2, F32), (2, 8, 8, 4, 8, 3, 3, 2, 2, 1, 1, F32), (1, 11, 6, 3, 4, 1, 1, 2, 2, 1, 1, F32),
                                  (2, 10, 10, 8, 8, 3, 3, 1, 1, 4, 4, F32), (3, 7, 9, 16, 8, 5, 3, 2, 1, 1, 2, BF16), (2, 16, 16, 32, 32, 3, 3, 1, 1, 1, 1, BF16)])
def test_general_conv_fwd_dgrad_wgrad(L, case):
    B, H, W, Ci, Co, kh, kw, sh, sw, dh, dw, dt = case
    x = RNG.standard_normal((B, H, W, Ci))
    w = RNG.standard_normal((kh, kw, Ci, Co)) / np.sqrt(kh * kw * Ci)
    b = RNG.standard_normal(Co) * 0.3
    xr = rounded(x, dt).requires_grad_(True)
    wr = torch.as_tensor(w, dtype=torch.float32).double().requires_grad_(True)
    yr = T.relu(T.conv2d_general_same(xr, wr, (sh, sw), (dh, dw)) + torch.as_tensor(b, dtype=torch.float32).double())
    ho, wo = ctypes.c_int(), ctypes.c_int()
    L.gconv2d_out_size(H, W, sh, sw, ctypes.byref(ho), ctypes.byref(wo))
    assert (ho.value, wo.value) == tuple(yr.shape[1:3])
    xd, wd, bd = dev(x, dt), dev(w), dev(b)
    y = torch.empty(B, ho.value, wo.value, Co, dtype=tdt(dt)).cuda()
    geo = (B, H, W, Ci, Co, kh, kw, sh, sw, dh, dw)
    L.gconv2d_fwd(xd.data_ptr(), dt, wd.data_ptr(), bd.data_ptr(), y.data_ptr(), dt, *geo, 1, S())
    tol = 2e-5 if dt == F32 else 8e-3
    close(y.float().cpu().numpy(), yr.detach().numpy(), tol, "fwd")
    dy = RNG.standard_normal(tuple(yr.shape))
    pre = T.conv2d_general_same(xr, wr, (sh, sw), (dh, dw))              # gradients of the un-activated convolution
    (pre * rounded(dy, dt)).sum().backward()
    dyd = dev(dy, dt)
    dx = torch.empty_like(xd)
    L.gconv2d_dgrad(dyd.data_ptr(), dt, wd.data_ptr(), dx.data_ptr(), dt, *geo, S())
    close(dx.float().cpu().numpy(), xr.grad.numpy(), tol, "dgrad")
    dwd = torch.full((kh, kw, Ci, Co), 0.25, dtype=torch.float32).cuda()
    L.gconv2d_wgrad(xd.data_ptr(), dt, dyd.data_ptr(), dt, dwd.data_ptr(), *geo, S())
    close(dwd.cpu().numpy() - 0.25, wr.grad.numpy(), 3e-5 if dt == F32 else 2e-5, "wgrad (accumulates)")


@pytest.mark.parametrize("case", [(2, 8, 8, 16, F32), (3, 7, 5, 6, F32), (2, 1, 9, 8, BF16), (2, 16, 16, 32, BF16)])
def test_maxpool_fwd_bwd(L, case):
    B, H, W, C, dt = case
    x = RNG.standard_normal((B, H, W, C))
    x[0, :2, :2, 0] = 0.0                                               # a window of ties (ReLU zeros): first element wins
    xr = rounded(x, dt).requires_grad_(True)
    yr = T.max_pool_2x2_same(xr)
    xd = dev(x, dt)
    y = torch.empty(B, (H + 1) // 2, (W + 1) // 2, C, dtype=tdt(dt)).cuda()
    L.maxpool2x2_fwd(xd.data_ptr(), dt, y.data_ptr(), B, H, W, C, S())
    close(y.float().cpu().numpy(), yr.detach().numpy(), 1e-7, "maxpool fwd")
    dy = RNG.standard_normal(tuple(yr.shape))
    (yr * rounded(dy, dt)).sum().backward()
    dyd = dev(dy, dt)
    dx = torch.full_like(xd, 7.0)
    L.maxpool2x2_bwd(xd.data_ptr(), dyd.data_ptr(), dt, dx.data_ptr(), B, H, W, C, S())
    close(dx.float().cpu().numpy(), xr.grad.numpy(), 1e-7, "maxpool bwd")


def test_spatial_window_pad_and_crop(L):
    x = RNG.standard_normal((2, 6, 5, 3))
    xd = dev(x)
    for (oh, ow, oy, ox) in [(9, 8, -1, -1), (4, 3, 1, 1), (6, 5, 0, 0), (7, 4, -1, 1)]:
        out = torch.full((2, oh, ow, 3), 9.0).cuda()
        L.spatial_window(xd.data_ptr(), out.data_ptr(), F32, 2, 6, 5, oh, ow, 3, oy, ox, S())
        want = T.spatial_window(torch.as_tensor(x, dtype=torch.float32), oh, ow, oy, ox)
        assert torch.equal(out.cpu(), want), (oh, ow, oy, ox)


def test_dropout_follows_the_philox_contract(L):
    B, shape = 3, (3, 5, 7, 6)
    x = RNG.standard_normal(shape)
    xd = dev(x)
    step = torch.tensor([4], dtype=torch.int32).cuda()
    y = torch.empty_like(xd)
    L.dropout(xd.data_ptr(), y.data_ptr(), F32, int(np.prod(shape[1:])), B, 0.7, 1234567890123, step.data_ptr(), 99, 5, S())
    keep = T.dropout_keep_mask(shape, 0.7, 1234567890123, 4, 99, sample_offset=5)
    want = np.where(keep, x.astype(np.float32) / np.float32(0.7), 0.0)
    close(y.cpu().numpy(), want, 1e-6, "dropout")
    assert 0.6 < keep.mean() < 0.8
    L.dropout(xd.data_ptr(), y.data_ptr(), F32, int(np.prod(shape[1:])), B, 1.0, 1, step.data_ptr(), 0, 0, S())
    assert torch.equal(y, xd)


@pytest.mark.parametrize("norm", ["identity", "batch_norm"])
def test_extra_layers_in_a_graph(norm):
    """The layer functions as code written against tfwrapper/layers.py calls them -- conv2D(strides=2, 5x5), dilated_conv2D(rate 2),
    maxpool2D, pad_to_size, crop_and_concat with a real crop, dropout, dense_layer -- compiled by the engine into one training
    plan: loss and every gradient against torch autograd of the oracle primitives (dropout mask from the Philox contract)."""
    from phiseg_code_amd import engine
    from phiseg_code_amd import graph as G
    from phiseg_code_amd.tfwrapper import activations as act
    from phiseg_code_amd.tfwrapper import layers
    from phiseg_code_amd.tfwrapper import normalisation as tfnorm
    B, H, C = 3, 12, 2
    g = G.reset_default_graph()
    x_inp = G.placeholder(G.KIND_F32, [None, H, H, 3], name="x_input")
    s_inp = G.placeholder(G.KIND_U8, [None, 6, 6], name="s_input")
    nfn = getattr(tfnorm, norm)
    with g.variable_scope("net"):
        a = layers.conv2D(x_inp, "c1", num_filters=8, kernel_size=(5, 5), strides=(2, 2), normalisation=nfn, training=True)     # 6 x 6
        d = layers.dilated_conv2D(x_inp, "dil", num_filters=4, rate=2, normalisation=nfn, training=True)                         # 12 x 12
        m = layers.maxpool2D(d)                                                                                                  # 6 x 6
        big = layers.pad_to_size(m, [None, 9, 8, 4])                                                                              # 9 x 8
        cat = layers.crop_and_concat([a, big])                                                                                   # crop back: 6 x 6, 12 ch
        dr = layers.dropout(cat, keep_prob=0.8, training=True)
        fc = layers.dense_layer(dr, "fc", hidden_units=5, normalisation=nfn, training=True)                                      # [B,1,1,5]
        gate = layers.conv2D(G.tile_pixels(G.global_average_pool(fc), 6, 6), "mix", num_filters=12, kernel_size=(1, 1),
                             normalisation=tfnorm.identity, training=True)
        s = layers.conv2D(G.concat([dr, gate], axis=-1), "head", num_filters=C, kernel_size=(1, 1), activation=act.identity)
    ce, _ = G.residual_multinoulli([s], s_inp, 1.0)
    loss = G.weighted_sum([ce[0]], [1.0])
    names = [n for n in g.variables]
    assert "net/fc/W" in names and g.variables["net/fc/W"].shape == (6 * 6 * 12, 5) and "net/dil/b" in names
    store = engine.ParamStore(g, seed=3)
    rng = np.random.default_rng(7)
    vals = {n: (v.initial_value(3) + (0.1 * rng.standard_normal(v.shape) if not n.endswith("/W") else 0)).astype(np.float32)
            for n, v in g.variables.items()}
    for n in vals:
        if n.endswith("moving_variance"):
            vals[n] = np.abs(vals[n]) + 0.5
    store.load(vals)
    seed = 4242
    plan = engine.Plan(store, [loss, s], loss=loss, batch=B, training=True, compute_dtype="f32", optimize=False, use_hip_graph=False,
                       rng_seed=seed)
    x = rng.standard_normal((B, H, H, 3)).astype(np.float32)
    lab = rng.integers(0, C, (B, 6, 6)).astype(np.uint8)
    plan.set_input("x_input", x)
    plan.set_input("s_input", lab)
    plan.run(sync=True)
    got_loss, got_s = float(plan.fetch(loss)), plan.fetch(s)
    grads = store.export(grads=True)
    # oracle
    p = {n: torch.as_tensor(v, dtype=torch.float64).requires_grad_(not n.rsplit("/", 1)[-1].startswith("moving_")) for n, v in vals.items()}

    def nrm(t, scope):
        if norm == "identity":
            return t
        y, _, _ = T.batch_norm_train(t, p[scope + "/batch_norm/BatchNorm/gamma"], p[scope + "/batch_norm/BatchNorm/beta"])
        return y
    xt = torch.as_tensor(x, dtype=torch.float64)
    a = T.conv2d_general_same(xt, p["net/c1/W"], (2, 2), (1, 1))
    if norm != "batch_norm":
        a = T.bias_add(a, p["net/c1/b"])
    a = T.relu(nrm(a, "net/c1"))
    d = T.relu(nrm(T.bias_add(T.conv2d_general_same(xt, p["net/dil/W"], (1, 1), (2, 2)), p["net/dil/b"]), "net/dil"))   # bias kept
    m = T.max_pool_2x2_same(d)
    big = torch.nn.functional.pad(m, (0, 0, 1, 1, 1, 2))                                  # (9 - 6) // 2 = 1 top, 2 bottom; 1 / 1 in x
    cat = torch.cat([a, big[:, 1:7, 1:7]], dim=-1)                                        # crop start (9 - 6) // 2 = 1, (8 - 6) // 2 = 1
    import zlib
    dname = [op.name for op in g.ops if op.type == "dropout"][0]
    keep = T.dropout_keep_mask(tuple(cat.shape), 0.8, seed, 0, zlib.crc32(dname.encode()) & 0x3FFFFFFF)
    dr = cat * torch.as_tensor(keep, dtype=torch.float64) / float(np.float32(0.8))
    fc = dr.reshape(B, -1) @ p["net/fc/W"] + p["net/fc/b"]
    fc = T.relu(nrm(fc.reshape(B, 1, 1, 5), "net/fc"))
    gate = T.relu(T.bias_add(T.conv2d_same(fc.mean(dim=(1, 2)).reshape(B, 1, 1, 5).expand(B, 6, 6, 5), p["net/mix/W"]), p["net/mix/b"]))
    so = T.bias_add(T.conv2d_same(torch.cat([dr, gate], dim=-1), p["net/head/W"]), p["net/head/b"])
    ref = T.multinoulli_loss_with_logits(T.one_hot(torch.as_tensor(lab), C, torch.float64), so)
    ref.backward()
    np.testing.assert_allclose(got_s, so.detach().numpy(), rtol=0, atol=3e-4 * float(so.abs().max()))
    np.testing.assert_allclose(got_loss, float(ref), rtol=3e-5)
    checked = 0
    for n, t in p.items():
        if t.grad is None:
            continue
        r = t.grad.numpy()
        if np.abs(r).max() < 1e-9:
            assert np.abs(grads[n]).max() < 1e-4, n
            continue
        np.testing.assert_allclose(grads[n], r, rtol=0, atol=3e-3 * max(np.abs(r).max(), 1e-6), err_msg=n)
        checked += 1
    assert checked >= 8
    # inference mode: dropout is the identity
    plan2 = engine.Plan(store, [s], loss=None, batch=B, training=False, compute_dtype="f32", use_hip_graph=False, rng_seed=seed)
    plan2.set_input("x_input", x)
    plan2.run(sync=True)
    assert np.isfinite(plan2.fetch(s)).all()


def _oracle_norm(p, t, scope, norm):
    if norm == "identity":
        return t
    if norm == "batch_norm":
        y, _, _ = T.batch_norm_train(t, p[scope + "/BatchNorm/gamma"], p[scope + "/BatchNorm/beta"])
        return y
    return T.group_norm(t, p[scope + "/gamma"], p[scope + "/beta"], 2)


def _oracle_skip(p, x, scope, cout, down, proj, norm):
    cin = x.shape[-1]
    if cin == cout and not down:
        return x
    st = (2, 2) if down else (1, 1)
    if proj:
        t = T.bias_add(T.conv2d_general_same(x, p[scope + "/projection/W"], st, (1, 1)), p[scope + "/projection/b"])
        return T.relu(_oracle_norm(p, t, scope + "/bn_projection", norm))
    pad = (cout - cin) // 2
    t = torch.nn.functional.pad(x, (pad, pad))
    return t[:, ::2, ::2, :] if down else t


@pytest.mark.parametrize("norm", ["identity", "batch_norm", "group_norm2D"])
def test_residual_units_in_a_graph(norm):
    """residual_unit2D (tfwrapper/layers.py:428-478) and identity_residual_unit2D (:481-536) with every skip variant -- plain,
    zero-padded channels + strided slice, 1x1 strided projection -- under identity / batch / group norm, as one training plan against
    torch autograd of the oracle primitives; variable names as the reference's scopes produce them."""
    from phiseg_code_amd import engine
    from phiseg_code_amd import graph as G
    from phiseg_code_amd.tfwrapper import activations as act
    from phiseg_code_amd.tfwrapper import layers
    from phiseg_code_amd.tfwrapper import normalisation as tfnorm
    B, H, C = 2, 8, 2
    g = G.reset_default_graph()
    x_inp = G.placeholder(G.KIND_F32, [None, H, H, 4], name="x_input")
    s_inp = G.placeholder(G.KIND_U8, [None, 2, 2], name="s_input")
    nfn = getattr(tfnorm, norm)
    kw = dict(normalisation=nfn, training=True, num_groups=2)
    with g.variable_scope("net"):
        r1 = layers.residual_unit2D(x_inp, "r1", num_filters=4, **kw)                                           # plain skip
        r2 = layers.residual_unit2D(r1, "r2", num_filters=8, down_sample=True, **kw)                            # pad + ::2      -> 4 x 4
        r3 = layers.residual_unit2D(r2, "r3", num_filters=6, projection=True, **kw)                             # projection
        i1 = layers.identity_residual_unit2D(r3, "i1", num_filters=6, **kw)                                     # plain skip
        i2 = layers.identity_residual_unit2D(i1, "i2", num_filters=12, down_sample=True, projection=False, **kw)  # pad + ::2   -> 2 x 2
        i3 = layers.identity_residual_unit2D(i2, "i3", num_filters=8, projection=True, **kw)
        s = layers.conv2D(i3, "head", num_filters=C, kernel_size=(1, 1), activation=act.identity)
    ce, _ = G.residual_multinoulli([s], s_inp, 1.0)
    loss = G.weighted_sum([ce[0]], [1.0])
    bn = {"identity": None, "batch_norm": "BatchNorm/gamma", "group_norm2D": "gamma"}[norm]
    for must in ("net/r1/conv1/W", "net/r1/conv1/b", "net/r3/projection/W", "net/i2/conv2/b") + \
            (("net/r1/bn1/" + bn, "net/r3/bn_projection/" + bn, "net/i1/bn2/" + bn) if bn else ()):
        assert must in g.variables, must
    store = engine.ParamStore(g, seed=3)
    rng = np.random.default_rng(7)
    vals = {n: (v.initial_value(3) + (0.1 * rng.standard_normal(v.shape) if not n.endswith("/W") else 0)).astype(np.float32)
            for n, v in g.variables.items()}
    for n in vals:
        if n.endswith("moving_variance"):
            vals[n] = np.abs(vals[n]) + 0.5
    store.load(vals)
    plan = engine.Plan(store, [loss, s], loss=loss, batch=B, training=True, compute_dtype="f32", optimize=False, use_hip_graph=False)
    x = rng.standard_normal((B, H, H, 4)).astype(np.float32)
    lab = rng.integers(0, C, (B, 2, 2)).astype(np.uint8)
    plan.set_input("x_input", x)
    plan.set_input("s_input", lab)
    plan.run(sync=True)
    got_loss, got_s = float(plan.fetch(loss)), plan.fetch(s)
    grads = store.export(grads=True)
    p = {n: torch.as_tensor(v, dtype=torch.float64).requires_grad_(not n.rsplit("/", 1)[-1].startswith("moving_")) for n, v in vals.items()}

    def conv(t, scope, st=(1, 1)):
        return T.bias_add(T.conv2d_general_same(t, p[scope + "/W"], st, (1, 1)), p[scope + "/b"])

    def res(t, scope, cout, down=False, proj=False):
        st = (2, 2) if down else (1, 1)
        c1 = T.relu(_oracle_norm(p, conv(t, scope + "/conv1", st), scope + "/bn1", norm))
        c2 = _oracle_norm(p, conv(c1, scope + "/conv2"), scope + "/bn2", norm)
        return T.relu(_oracle_skip(p, t, scope, cout, down, proj, norm) + c2)

    def ires(t, scope, cout, down=False, proj=True):
        st = (2, 2) if down else (1, 1)
        o1 = conv(T.relu(_oracle_norm(p, t, scope + "/bn1", norm)), scope + "/conv1", st)
        o2 = conv(T.relu(_oracle_norm(p, o1, scope + "/bn2", norm)), scope + "/conv2")
        return _oracle_skip(p, t, scope, cout, down, proj, norm) + o2
    xt = torch.as_tensor(x, dtype=torch.float64)
    t = res(xt, "net/r1", 4)
    t = res(t, "net/r2", 8, down=True)
    t = res(t, "net/r3", 6, proj=True)
    t = ires(t, "net/i1", 6)
    t = ires(t, "net/i2", 12, down=True, proj=False)
    t = ires(t, "net/i3", 8, proj=True)
    so = T.bias_add(T.conv2d_same(t, p["net/head/W"]), p["net/head/b"])
    ref = T.multinoulli_loss_with_logits(T.one_hot(torch.as_tensor(lab), C, torch.float64), so)
    ref.backward()
    np.testing.assert_allclose(got_s, so.detach().numpy(), rtol=0, atol=5e-4 * float(so.detach().abs().max()))
    np.testing.assert_allclose(got_loss, float(ref), rtol=5e-5)
    checked = 0
    for n, tt in p.items():
        if tt.grad is None:
            continue
        r = tt.grad.numpy()
        if np.abs(r).max() < 1e-9:
            assert np.abs(grads[n]).max() < 2e-4, n
            continue
        np.testing.assert_allclose(grads[n], r, rtol=0, atol=5e-3 * max(np.abs(r).max(), 1e-6), err_msg=n)
        checked += 1
    assert checked >= 20


def test_reshape_pool_layer():
    """reshape_pool2D_layer (layers.py:57-67): space-to-depth by strided slices, forward and gradient through the engine."""
    from phiseg_code_amd import engine
    from phiseg_code_amd import graph as G
    from phiseg_code_amd.tfwrapper import activations as act
    from phiseg_code_amd.tfwrapper import layers
    B, H, W, C = 2, 4, 6, 3
    g = G.reset_default_graph()
    x_inp = G.placeholder(G.KIND_F32, [None, H, W, C], name="x_input")
    s_inp = G.placeholder(G.KIND_U8, [None, 2, 2], name="s_input")          # (unused by the check; the loss needs labels)
    with g.variable_scope("net"):
        c0 = layers.conv2D(x_inp, "c0", num_filters=C, kernel_size=(1, 1), activation=act.identity)
        rp = layers.reshape_pool2D_layer(c0)
    assert rp.get_shape().as_list()[1:] == [2, 3, 12]
    store = engine.ParamStore(g, seed=1)
    vals = {"net/c0/W": np.eye(C, dtype=np.float32).reshape(1, 1, C, C), "net/c0/b": np.zeros(C, dtype=np.float32)}
    store.load(vals)
    plan = engine.Plan(store, [rp], loss=None, batch=B, training=False, compute_dtype="f32", use_hip_graph=False)
    x = RNG.standard_normal((B, H, W, C)).astype(np.float32)
    plan.set_input("x_input", x)
    plan.run(sync=True)
    want = np.concatenate([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], axis=3)
    np.testing.assert_allclose(plan.fetch(rp), want, rtol=0, atol=1e-6)


def test_window4_add_act_and_per_sample_partial_reduction(L):
    """phx_window4_fwd / _bwd (strided window with a channel offset), phx_add_act and phx_norm_reduce_partials_ns through the C ABI."""
    x = RNG.standard_normal((2, 7, 6, 4)).astype(np.float32)
    xd = dev(x)
    # zero-padded channels (1 in front, 1 behind) + [:, ::2, ::2, :]
    out = torch.full((2, 4, 3, 6), 9.0).cuda()
    L.window4_fwd(xd.data_ptr(), out.data_ptr(), F32, 2, 7, 6, 4, 4, 3, 6, 2, 2, 0, 0, -1, S())
    want = np.pad(x, ((0, 0), (0, 0), (0, 0), (1, 1)))[:, ::2, ::2, :]
    np.testing.assert_array_equal(out.cpu().numpy(), want)
    d = RNG.standard_normal((2, 4, 3, 6)).astype(np.float32)
    dd, dx = dev(d), torch.full((2, 7, 6, 4), 9.0).cuda()
    L.window4_bwd(dd.data_ptr(), dx.data_ptr(), F32, 2, 7, 6, 4, 4, 3, 6, 2, 2, 0, 0, -1, S())
    ref = np.zeros_like(x)
    ref[:, ::2, ::2, :] = d[..., 1:5]
    np.testing.assert_array_equal(dx.cpu().numpy(), ref)
    # act(a + b), bf16
    a, b = torch.randn(1000, device="cuda").to(torch.bfloat16), torch.randn(1000, device="cuda").to(torch.bfloat16)
    y = torch.empty_like(a)
    L.add_act(a.data_ptr(), b.data_ptr(), y.data_ptr(), BF16, 1000, 1, S())
    assert torch.equal(y, torch.relu(a.float() + b.float()).to(torch.bfloat16))
    # per-sample reduction of per-tile partial sums: partial[ns * T + t][2][C] -> sums[ns][c][2]
    NS, Tt, C = 5, 7, 24
    part = RNG.standard_normal((NS * Tt, 2, C)).astype(np.float32)
    pd = dev(part)
    sums = torch.full((NS, C, 2), 3.0).cuda()
    L.norm_reduce_partials_ns(pd.data_ptr(), Tt, NS, C, sums.data_ptr(), S())
    want = part.reshape(NS, Tt, 2, C).sum(axis=1).transpose(0, 2, 1)
    close(sums.cpu().numpy(), want, 1e-6, "per-sample partial reduction (overwrites)")
