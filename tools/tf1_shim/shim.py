"""A numeric stand-in for the ~50 ``tensorflow`` 1.12 symbols the reference's hot-path files touch.

BUILD-CONTAINER-ONLY TOOL (never imported by the product, never needed on the GPU box).
``install()`` registers modules named ``tensorflow``, ``tensorflow.contrib.layers``,
``tensorflow.python.pywrap_tensorflow`` (+ empty ``nibabel`` / ``skimage`` / ``medpy`` stubs) in
``sys.modules`` so that the reference's OWN files
    /root/reference/tfwrapper/{layers,normalisation,utils}.py
    /root/reference/phiseg/model_zoo/{posteriors,priors,likelihoods}.py
    /root/reference/phiseg/phiseg_model.py   (loss methods only)
can be imported UNMODIFIED and executed eagerly; every primitive dispatches to
``oracle.tf1_ops``.  ``tools/make_goldens.py`` uses this to write ``tests/golden/*.npz`` in which
layer order, names, kernel sizes, bias / norm placement, concat order and teacher forcing are the
reference's code -- not our reading of it.  Nothing from /root/reference is copied.
"""
import contextlib
import sys
import types

import numpy as np
import torch

from oracle import tf1_ops as T


class State:
    dtype = torch.float64
    scope = []                 # variable-scope name stack
    variables = {}             # full name -> torch tensor
    var_order = []             # creation order (name, shape)
    provider = None            # fn(name, shape, init) -> numpy array
    eps_provider = None        # fn(scope_str, call_index, shape) -> numpy array
    eps_calls = {}             # scope_str -> count
    training = True
    moving_updates = {}
    conv_log = []              # (scope, kh, kw, cin, cout, h, w)
    collections = {}


S = State


def reset(dtype=torch.float64):
    S.dtype = dtype
    S.scope = []
    S.variables = {}
    S.var_order = []
    S.eps_calls = {}
    S.moving_updates = {}
    S.conv_log = []
    S.collections = {}


class _Op:
    def __init__(self, name):
        self.name = name


class TT:
    """Eager tensor wrapper exposing the few TF Tensor methods the reference uses."""

    def __init__(self, v, name="t"):
        self.v = v
        self.name = name + ":0"
        self.op = _Op(name)

    def get_shape(self):
        return _Shape(list(self.v.shape))

    @property
    def shape(self):
        return _Shape(list(self.v.shape))

    def __bool__(self):          # layers.py:675 does ``if biases:``
        return True

    def _b(self, o, f):
        ov = o.v if isinstance(o, TT) else o
        return TT(f(self.v, ov))

    def __add__(self, o): return self._b(o, lambda a, b: a + b)
    def __radd__(self, o): return self._b(o, lambda a, b: b + a)
    def __sub__(self, o): return self._b(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._b(o, lambda a, b: b - a)
    def __mul__(self, o): return self._b(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._b(o, lambda a, b: b * a)
    def __truediv__(self, o): return self._b(o, lambda a, b: a / b)
    def __neg__(self): return TT(-self.v)


class _Dim(int):
    @property
    def value(self):
        return int(self)


class _Shape(list):
    def as_list(self):
        return [int(d) for d in self]

    def __getitem__(self, i):
        r = list.__getitem__(self, i)
        return _Dim(r) if isinstance(r, int) else r


def _v(x):
    return x.v if isinstance(x, TT) else x


def _scope_str():
    return "/".join(S.scope)


class _VarScope:
    def __init__(self, name):
        self.name = name

    def reuse_variables(self):
        pass


@contextlib.contextmanager
def variable_scope(name, *a, **k):
    S.scope.append(name)
    try:
        yield _VarScope(_scope_str())
    finally:
        S.scope.pop()


def get_variable(name, shape=None, initializer=None, **kw):
    full = _scope_str() + "/" + name
    if full in S.variables:
        return S.variables[full]
    if shape is None:                        # get_bias_variable passes a tensor initializer
        init_t = _v(initializer)
        shape = list(init_t.shape)
        init = ("constant", float(init_t.reshape(-1)[0]) if init_t.numel() else 0.0)
    else:
        shape = [int(s) for s in shape]
        init = initializer if isinstance(initializer, tuple) else ("constant", 0.0)
    val = S.provider(full, shape, init)
    t = TT(torch.as_tensor(np.asarray(val), dtype=S.dtype).reshape(shape).clone().requires_grad_(True), full)
    S.variables[full] = t
    S.var_order.append((full, shape))
    return t


def Variable(initial, **kw):
    raise NotImplementedError("unnamed variables are not used on the hot path")


def constant(value, shape=None, dtype=None, **kw):
    t = torch.as_tensor(value, dtype=S.dtype)
    if shape is not None:
        t = t.expand(*shape).clone() if t.dim() == 0 else t.reshape(shape)
    return TT(t)


def constant_initializer(value=0.0, **kw):
    return ("constant", float(value))


def random_normal_initializer(mean=0.0, stddev=1.0, **kw):
    return ("normal", float(mean), float(stddev))


def variance_scaling_initializer(factor=2.0, mode="FAN_IN", uniform=False, **kw):
    return ("variance_scaling", factor, mode, uniform)


def xavier_initializer(uniform=True, **kw):
    return ("xavier", uniform)


def add_to_collection(name, v):
    S.collections.setdefault(name, []).append(v)


def get_collection(name):
    return S.collections.get(name, [])


def identity(x, name=None):
    return TT(_v(x))


def concat(values, axis, name=None):
    return TT(torch.cat([_v(v) for v in values], dim=axis))


def shape(x):
    return list(_v(x).shape)


def stack(vals, axis=0):
    return [int(_v(v)) if not isinstance(v, int) else v for v in vals]


def reshape(x, shp):
    shp = [int(_v(s)) for s in shp]
    return TT(_v(x).reshape(shp))


def tile(x, multiples):
    return TT(_v(x).repeat(*[int(m) for m in multiples]))


def tf_slice(x, begin, size):
    xv = _v(x)
    idx = []
    for d, (b, s) in enumerate(zip(begin, size)):
        b = int(b)
        idx.append(slice(b, None) if int(s) == -1 else slice(b, b + int(s)))
    return TT(xv[tuple(idx)])


def random_normal(shp, mean=0.0, stddev=1.0, dtype=None, **kw):
    sc = S.scope[0] if S.scope else ""
    k = S.eps_calls.get(sc, 0)
    S.eps_calls[sc] = k + 1
    shp = [int(s) for s in shp]
    e = S.eps_provider(sc, k, shp)
    return TT(torch.as_tensor(np.asarray(e), dtype=S.dtype).reshape(shp) * stddev + mean)


def reduce_mean(x, axis=None, keepdims=False, name=None, keep_dims=None):
    xv = _v(x)
    kd = keepdims if keep_dims is None else keep_dims
    if axis is None:
        return TT(xv.mean(), name or "mean")
    return TT(xv.mean(dim=tuple(axis) if isinstance(axis, (list, tuple)) else axis, keepdim=kd), name or "mean")


def reduce_sum(x, axis=None, keepdims=False, name=None, input_tensor=None):
    xv = _v(x if input_tensor is None else input_tensor)
    if axis is None:
        return TT(xv.sum())
    return TT(xv.sum(dim=tuple(axis) if isinstance(axis, (list, tuple)) else axis, keepdim=keepdims))


def sqrt(x): return TT(torch.sqrt(_v(x)))
def rsqrt(x): return TT(torch.rsqrt(_v(x)))
def square(x): return TT(_v(x) ** 2)
def log(x): return TT(torch.log(_v(x)))
def divide(a, b): return TT(_v(a) / _v(b))
def one_hot(x, depth): return TT(T.one_hot(_v(x), depth, S.dtype))


# ---- tf.nn ----------------------------------------------------------------------------------
def nn_conv2d(x, filter=None, strides=None, padding="SAME", **kw):
    assert list(strides) == [1, 1, 1, 1] and padding == "SAME"
    xv, wv = _v(x), _v(filter)
    S.conv_log.append((_scope_str(), int(wv.shape[0]), int(wv.shape[1]), int(wv.shape[2]),
                       int(wv.shape[3]), int(xv.shape[1]), int(xv.shape[2])))
    return TT(T.conv2d_same(xv, wv), _scope_str() + "/Conv2D")


def nn_bias_add(x, b): return TT(T.bias_add(_v(x), _v(b)), _scope_str() + "/BiasAdd")


def nn_avg_pool(x, ksize, strides, padding):
    assert list(ksize) == [1, 2, 2, 1] and list(strides) == [1, 2, 2, 1] and padding == "SAME"
    return TT(T.avg_pool_2x2_same(_v(x)))


def nn_relu(x): return TT(T.relu(_v(x)), _scope_str() + "/Relu")
def nn_softplus(x): return TT(T.softplus(_v(x)), _scope_str() + "/Softplus")
def nn_softmax(x): return TT(torch.softmax(_v(x), dim=-1))


def nn_moments(x, axes, keep_dims=False):
    xv = _v(x)
    m = xv.mean(dim=tuple(axes), keepdim=True)
    var = ((xv - m) ** 2).mean(dim=tuple(axes), keepdim=True)
    if not keep_dims:
        m, var = m.squeeze(), var.squeeze()
    return TT(m), TT(var)


def nn_softmax_xent_v2(labels=None, logits=None, **kw):
    return TT(-(_v(labels) * torch.log_softmax(_v(logits), dim=-1)).sum(dim=-1))


# ---- tf.image -------------------------------------------------------------------------------
class ResizeMethod:
    BILINEAR = 0
    NEAREST_NEIGHBOR = 1


def resize_images(x, size, method=ResizeMethod.BILINEAR, align_corners=False):
    assert not align_corners
    oh, ow = int(_v(size[0])), int(_v(size[1]))
    if method == ResizeMethod.NEAREST_NEIGHBOR:
        return TT(T.resize_nearest(_v(x), oh, ow))
    return TT(T.resize_bilinear_legacy(_v(x), oh, ow))


# ---- tf.contrib.layers.batch_norm -----------------------------------------------------------
def contrib_batch_norm(inputs, decay=0.999, epsilon=0.001, is_training=True, center=True, scale=True, **kw):
    assert abs(epsilon - T.BN_EPS) < 1e-12 and abs(decay - T.BN_DECAY) < 1e-12 and center and scale
    c = int(_v(inputs).shape[-1])
    with variable_scope("BatchNorm"):
        beta = get_variable("beta", [c], ("constant", 0.0))
        gamma = get_variable("gamma", [c], ("constant", 1.0))
        mm = get_variable("moving_mean", [c], ("constant", 0.0))
        mv = get_variable("moving_variance", [c], ("constant", 1.0))
        pref = _scope_str() + "/"
    training = is_training if isinstance(is_training, bool) else S.training
    if training:
        y, mean, var_u = T.batch_norm_train(_v(inputs), gamma.v, beta.v)
        # The reference graph instantiates prior and likelihood twice over shared variables
        # (phiseg_model.py:48-98), so TF holds TWO unordered moving-average assign ops per shared
        # layer.  The build defines the update by the training-graph instance, which
        # is built first -> first write wins here.
        S.moving_updates.setdefault(pref + "moving_mean", T.batch_norm_moving_update(mm.v.detach(), mean.detach()))
        S.moving_updates.setdefault(pref + "moving_variance", T.batch_norm_moving_update(mv.v.detach(), var_u.detach()))
    else:
        y = T.batch_norm_infer(_v(inputs), gamma.v, beta.v, mm.v, mv.v)
    return TT(y, pref + "FusedBatchNorm")


# ---------------------------------------------------------------------------------------------
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    nn = _mod("tensorflow.nn", conv2d=nn_conv2d, bias_add=nn_bias_add, avg_pool=nn_avg_pool,
              relu=nn_relu, softplus=nn_softplus, softmax=nn_softmax, moments=nn_moments,
              softmax_cross_entropy_with_logits_v2=nn_softmax_xent_v2,
              l2_loss=lambda x: TT((_v(x) ** 2).sum() / 2))
    image = _mod("tensorflow.image", resize_images=resize_images, ResizeMethod=ResizeMethod)
    summary = _mod("tensorflow.summary", histogram=lambda *a, **k: None, scalar=lambda *a, **k: None,
                   image=lambda *a, **k: None)
    cl = _mod("tensorflow.contrib.layers", batch_norm=contrib_batch_norm,
              variance_scaling_initializer=variance_scaling_initializer,
              xavier_initializer=xavier_initializer)
    contrib = _mod("tensorflow.contrib", layers=cl)
    pw = _mod("tensorflow.python.pywrap_tensorflow")
    py = _mod("tensorflow.python", pywrap_tensorflow=pw)

    class _Opt:            # tf.train.AdamOptimizer etc. are only *named* by the experiment configs
        def __init__(self, *a, **k): pass
    train = _mod("tensorflow.train", AdamOptimizer=type("AdamOptimizer", (_Opt,), {}),
                 MomentumOptimizer=type("MomentumOptimizer", (_Opt,), {}))
    _mod("tensorflow", nn=nn, image=image, summary=summary, contrib=contrib, python=py, train=train,
         variable_scope=variable_scope, get_variable=get_variable, Variable=Variable,
         constant=constant, constant_initializer=constant_initializer,
         random_normal_initializer=random_normal_initializer, add_to_collection=add_to_collection,
         get_collection=get_collection, identity=identity, concat=concat, shape=shape, stack=stack,
         reshape=reshape, tile=tile, slice=tf_slice, random_normal=random_normal,
         reduce_mean=reduce_mean, reduce_sum=reduce_sum, sqrt=sqrt, rsqrt=rsqrt, square=square,
         log=log, divide=divide, one_hot=one_hot, float32="float32", uint8="uint8", bool="bool")
    # absent third-party imports of the reference's utils.py / phiseg_model.py (never called here)
    _mod("nibabel")
    sk_m = _mod("skimage.measure")
    sk_t = _mod("skimage.transform")
    _mod("skimage", measure=sk_m, transform=sk_t)
    mm = _mod("medpy.metric", jc=None, dc=None)
    _mod("medpy", metric=mm)
