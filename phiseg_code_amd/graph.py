"""Symbolic graph of the PHiSeg step (the role TF1's graph mode plays in the reference).

The reference builds a static TensorFlow graph once (phiseg/phiseg_model.py:20-157) and then runs it
with ``sess.run``.  This module is the build-time half of the replacement: ``tfwrapper.layers`` /
``phiseg.model_zoo`` (our own code, same call signatures) create ``Op`` nodes over ``Tensor`` handles with
static NHWC shapes (batch dimension ``None``), variables live under nested ``variable_scope`` names that
match the reference's TF variable names.  ``engine.Plan`` then lowers a set of
fetches, for a concrete batch size and training flag, to a fixed list of HIP kernel launches that is
captured into a hipGraph -- HIP streams and graphs instead of a tracing compiler.
"""
import contextlib
import zlib
from collections import OrderedDict

import numpy as np

KIND_ACT, KIND_F32, KIND_U8 = "act", "f32", "u8"     # "act" = stored in the plan's compute dtype


class _Shape(list):
    def as_list(self):
        return list(self)


class Tensor:
    def __init__(self, op, shape, kind=KIND_ACT, name=None):
        self.op = op
        self.shape = tuple(shape)
        self.kind = kind
        self.name = name or (op.name if op is not None else "t")
        self.consumers = []
        self.bmul = 1                   # rows per fed image: > 1 behind tile_batch (n Monte-Carlo samples per image)

    def get_shape(self):
        return _Shape(self.shape)

    # the few arithmetic forms the zoo uses: s_oh - 0.5 ; sigma * eps ; mu + (sigma * eps)
    def __sub__(self, c):
        if not isinstance(c, (int, float)):
            raise NotImplementedError("only tensor - constant is part of the hot path")
        return get_default_graph().add_op("sub_const", [self], dict(c=float(c)), [(self.shape, self.kind)])[0]

    def __mul__(self, o):
        if not isinstance(o, Tensor) or o.shape != self.shape:
            raise NotImplementedError("only same-shape tensor * tensor is part of the hot path")
        return get_default_graph().add_op("mul", [self, o], {}, [(self.shape, KIND_F32)])[0]

    def __add__(self, o):
        if not isinstance(o, Tensor) or o.shape != self.shape:
            raise NotImplementedError("only same-shape tensor + tensor is part of the hot path")
        return get_default_graph().add_op("add", [self, o], {}, [(self.shape, KIND_F32)])[0]

    def __repr__(self):
        return "<Tensor %s %s %s>" % (self.name, self.shape, self.kind)


class Op:
    def __init__(self, type_, inputs, attrs, name):
        self.type = type_
        self.inputs = list(inputs)
        self.attrs = attrs
        self.name = name
        self.outputs = []


class Variable:
    def __init__(self, name, shape, initializer, trainable=True):
        self.name = name
        self.shape = tuple(int(s) for s in shape)
        self.initializer = initializer
        self.trainable = trainable

    @property
    def size(self):
        return int(np.prod(self.shape))

    def initial_value(self, seed=0):
        """Value at initialisation: a function of (seed, variable name, shape) only -- the Philox stream contract of
        phiseg_code_amd/philox_host.py -- so every replica, every run and the CPU oracle start from identical weights."""
        from phiseg_code_amd import philox_host
        rng = philox_host.VariableStream(seed, self.name)
        return np.asarray(self.initializer(self.shape, rng), dtype=np.float32).reshape(self.shape)

    def __bool__(self):
        return True


class _ScopeHandle:
    def __init__(self, graph, name):
        self.graph, self.name = graph, name

    def reuse_variables(self):
        self.graph._reuse[self.name] = True


class Graph:
    def __init__(self):
        self.ops = []
        self.variables = OrderedDict()
        self._scope = []
        self._reuse = {}
        self.collections = {}
        self._names = {}

    # ---- scopes / variables (tf.variable_scope / tf.get_variable) --------------------------------
    def scope_name(self):
        return "/".join(self._scope)

    @contextlib.contextmanager
    def variable_scope(self, name):
        self._scope.append(name)
        full = self.scope_name()
        try:
            yield _ScopeHandle(self, full)
        finally:
            self._scope.pop()

    def _reusing(self):
        return any(self._reuse.get("/".join(self._scope[:i + 1]), False) for i in range(len(self._scope)))

    def get_variable(self, name, shape, initializer, trainable=True):
        full = (self.scope_name() + "/" if self._scope else "") + name
        if full in self.variables:
            if not self._reusing():
                raise ValueError("Variable %s already exists (enter the scope with scope_reuse=True)" % full)
            v = self.variables[full]
            if tuple(shape) != v.shape:
                raise ValueError("Variable %s re-used with shape %s != %s" % (full, tuple(shape), v.shape))
            return v
        if self._reusing():
            raise ValueError("Variable %s does not exist but the scope is in reuse mode" % full)
        v = Variable(full, shape, initializer, trainable)
        self.variables[full] = v
        return v

    def add_to_collection(self, name, v):
        self.collections.setdefault(name, []).append(v)

    def get_collection(self, name):
        return list(self.collections.get(name, []))

    # ---- ops -------------------------------------------------------------------------------------
    def unique_name(self, base):
        k = self._names.get(base, 0)
        self._names[base] = k + 1
        return base if k == 0 else "%s_%d" % (base, k)

    def add_op(self, type_, inputs, attrs, out_specs, name=None):
        op = Op(type_, inputs, attrs, self.unique_name((self.scope_name() + "/" if self._scope else "") + (name or type_)))
        for i, (shape, kind) in enumerate(out_specs):
            op.outputs.append(Tensor(op, shape, kind, op.name + (":%d" % i if len(out_specs) > 1 else "")))
        bm = {t.bmul for t in inputs if t.shape and t.shape[0] is None}
        if len(bm) > 1:
            raise ValueError("op %s mixes tensors with %s rows per image (tile_batch the smaller one first)" % (op.name, sorted(bm)))
        rows = (max(bm) if bm else 1) * (int(attrs.get("tile", 1)) if type_ == "tile_batch" else 1)
        for o in op.outputs:
            o.bmul = rows
        for t in inputs:
            t.consumers.append(op)
        self.ops.append(op)
        return op.outputs


_default = Graph()


def get_default_graph():
    return _default


def set_default_graph(g):
    """Make `g` the graph new ops are added to (a model adding a graph instance after construction)."""
    global _default
    _default = g
    return g


def reset_default_graph():
    global _default
    _default = Graph()
    return _default


def variable_scope(name):
    return get_default_graph().variable_scope(name)


# --------------------------------------------------------------------------------------------------
# op constructors (the symbols our tfwrapper / model_zoo / phiseg_model are written against)
def placeholder(kind, shape, name):
    return get_default_graph().add_op("placeholder", [], dict(), [(tuple(shape), kind)], name=name)[0]


def constant(value):
    """tf.constant(scalar): only ever a placeholder value (the dummy posterior / prior of the deterministic baseline)."""
    return get_default_graph().add_op("constant", [], dict(value=float(value)), [((), KIND_F32)], name="const")[0]


def one_hot(s, depth):
    """tf.one_hot (phiseg_model.py:29): [.., H, W] u8 -> [.., H, W, depth]."""
    return get_default_graph().add_op("one_hot", [s], dict(depth=int(depth)), [(s.shape + (int(depth),), KIND_ACT)])[0]


def concat(values, axis=-1, name=None):
    """tf.concat along the channel axis of exactly two NHWC tensors (all the zoo needs)."""
    if len(values) != 2:
        raise NotImplementedError("concat of exactly two tensors")
    a, b = values
    if axis not in (-1, 3) or a.shape[:-1] != b.shape[:-1]:
        raise ValueError("concat: channel axis only, equal spatial shapes (%s vs %s)" % (a.shape, b.shape))
    kind = KIND_F32 if (a.kind == KIND_F32 and b.kind == KIND_F32) else KIND_ACT
    return get_default_graph().add_op("concat", [a, b], {}, [(a.shape[:-1] + (a.shape[-1] + b.shape[-1],), kind)],
                                      name=name or "concat")[0]


def random_normal(like, stream):
    """tf.random_normal(tf.shape(like), 0, 1): the Philox stream id fixes the noise contract
    (oracle/philox.py): stream = 16 * net + level."""
    return get_default_graph().add_op("random_normal", [like], dict(stream=int(stream)), [(like.shape, KIND_F32)])[0]


def conv_unit(x, W, b, ksize, norm, norm_vars, act, training, num_groups=None, head=False, name="conv", transposed=None,
              general=None):
    """conv (SAME, stride 1) -> [+bias] -> [norm] -> act as ONE node (tfwrapper/layers.py:122-135).
    transposed = (kh, kw, sh, sw): tf.nn.conv2d_transpose instead (layers.py:197-258), W is [kh, kw, Cout, Cin] and the output
    is sh x sw times larger.  general = (kh, kw, sh, sw, dh, dw): strided / dilated SAME convolution (conv2D with strides,
    dilated_conv2D, dense_layer as a 1x1 convolution of the flattened input) on the direct kernels of csrc/gconv.hip; the
    output is ceil(H / sh) x ceil(W / sw)."""
    kind = KIND_F32 if head else KIND_ACT
    attrs = dict(W=W, b=b, ksize=int(ksize), norm=norm, norm_vars=norm_vars, act=act, training=training,
                 num_groups=num_groups, head=head, transposed=transposed, general=general)
    if transposed is not None:
        kh, kw, sh, sw = transposed
        shape = (x.shape[0], x.shape[1] * sh, x.shape[2] * sw, W.shape[2])
    elif general is not None:
        sh, sw = general[2], general[3]
        shape = (x.shape[0], -(-x.shape[1] // sh), -(-x.shape[2] // sw), W.shape[-1])
    else:
        shape = x.shape[:3] + (W.shape[3],)
    return get_default_graph().add_op("conv_unit", [x], attrs, [(shape, kind)], name=name)[0]


def avg_pool2x2(x):
    n, h, w, c = x.shape
    return get_default_graph().add_op("avgpool", [x], {}, [((n, (h + 1) // 2, (w + 1) // 2, c), x.kind)])[0]


def max_pool2x2(x):
    """tf.nn.max_pool 2x2 / stride 2 / SAME (tfwrapper/layers.py:18-28)."""
    n, h, w, c = x.shape
    return get_default_graph().add_op("maxpool", [x], {}, [((n, (h + 1) // 2, (w + 1) // 2, c), x.kind)])[0]


def spatial_window(x, out_h, out_w, off_y, off_x, name="window"):
    """out[b, y, x] = x[b, y + off_y, x + off_x] inside x, 0 outside: zero padding (pad_to_size, layers.py:625-650) for negative
    offsets, centre crop (crop_and_concat, layers.py:586-622) for positive ones."""
    n, h, w, c = x.shape
    return get_default_graph().add_op("spatial_window", [x], dict(off=(int(off_y), int(off_x))),
                                      [((n, int(out_h), int(out_w), c), x.kind)], name=name)[0]


def dropout(x, keep_prob, training, name="dropout"):
    """tf.nn.dropout(x, keep_prob) in training mode, identity otherwise (layers.py:653-668); the keep mask follows the Philox
    contract with stream id crc32(op name) (oracle.tf1_ops.dropout_keep_mask)."""
    import zlib
    out = get_default_graph().add_op("dropout", [x], dict(keep_prob=float(keep_prob), training=training), [(x.shape, x.kind)],
                                     name=name)[0]
    out.op.attrs["stream"] = zlib.crc32(out.op.name.encode()) & 0x3FFFFFFF
    return out


def window4(x, out_h, out_w, out_c, stride=(1, 1), off=(0, 0, 0), name="window"):
    """out[b, y, x, c] = x[b, y sy + oy, x sx + ox, c + oc] inside x, 0 outside: tf.pad along the channel axis (oc = -pad) and the
    strided slice identity[:, ::2, ::2, :] of the residual units' skip path (tfwrapper/layers.py:465-470)."""
    n = x.shape[0]
    return get_default_graph().add_op("window4", [x], dict(stride=(int(stride[0]), int(stride[1])), off=tuple(int(v) for v in off)),
                                      [((n, int(out_h), int(out_w), int(out_c)), x.kind)], name=name)[0]


def add_act(a, b, act="identity", name="add"):
    """act(a + b): tf.add(skip, conv2) [+ activation] of the residual units (layers.py:474-475, 534)."""
    if tuple(a.shape) != tuple(b.shape):
        raise ValueError("add: shapes %s and %s differ" % (a.shape, b.shape))
    return get_default_graph().add_op("add_act", [a, b], dict(act=act), [(a.shape, a.kind)], name=name)[0]


def norm_act(x, norm, norm_vars, act, training, num_groups=None, name="norm"):
    """act(normalisation(x)) as a node of its own: the pre-activation order of identity_residual_unit2D (layers.py:512-518), where
    the normalisation does not follow a convolution."""
    attrs = dict(norm=norm, norm_vars=norm_vars, act=act, training=training, num_groups=num_groups)
    return get_default_graph().add_op("norm_act", [x], attrs, [(x.shape, x.kind)], name=name)[0]


def flatten(x):
    """tfwrapper/utils.py:16-22 flatten: [B, ...] -> [B, 1, 1, prod(...)] (NHWC memory order kept: a view, no launch); the
    extra unit axes keep every tensor of the engine four-dimensional."""
    n = x.shape[0]
    f = int(np.prod(x.shape[1:]))
    return get_default_graph().add_op("flatten", [x], {}, [((n, 1, 1, f), x.kind)], name="flatten")[0]


def bilinear_up2x(x, name="ups"):
    n, h, w, c = x.shape
    return get_default_graph().add_op("bilinear_up", [x], {}, [((n, 2 * h, 2 * w, c), x.kind)], name=name)[0]


def resize_nearest(x, out_hw):
    """tf.image.resize_images(..., NEAREST_NEIGHBOR) by an integer power-of-two factor (likelihoods.py:221).
    Never materialised: the loss / aggregation kernels read the coarse logits through a shift."""
    n, h, w, c = x.shape
    f = out_hw[0] // h
    if h * f != out_hw[0] or w * f != out_hw[1] or f & (f - 1) or f > 16:
        raise ValueError("nearest resize: integer power-of-two factor <= 16 required (%s -> %s)" % ((h, w), out_hw))
    return get_default_graph().add_op("nn_resize", [x], dict(shift=f.bit_length() - 1),
                                      [((n, out_hw[0], out_hw[1], c), KIND_F32)])[0]


def tile_batch(x, n):
    """[B, ...] -> [B * n, ...], row b * n + k = x[b]: every image's feature map repeated for its n Monte-Carlo samples."""
    if n == 1:
        return x
    return get_default_graph().add_op("tile_batch", [x], dict(tile=int(n)), [(x.shape, x.kind)], name="tile_batch")[0]


def global_average_pool(x, name=None):
    n, h, w, c = x.shape
    return get_default_graph().add_op("global_avgpool", [x], {}, [((n, c), KIND_F32)], name=name or "gap")[0]


def tile_pixels(z, h, w):
    """tf.reshape(z,[bs,1,1,zdim]) + tf.tile (likelihoods.py:147-149): [B, C] -> [B, h, w, C]."""
    n, c = z.shape
    return get_default_graph().add_op("tile_pixels", [z], {}, [((n, h, w, c), KIND_ACT)])[0]


def residual_multinoulli(s_list, labels, weight):
    """phiseg_model.py:229-262 for all levels at once -> (list of per-level loss scalars, s_accum[0])."""
    g = get_default_graph()
    L = len(s_list)
    outs = g.add_op("residual_ce", list(s_list) + [labels], dict(L=L, weight=float(weight)),
                    [((), KIND_F32)] * L + [(s_list[0].shape, KIND_F32)], name="residual_multinoulli")
    return outs[:L], outs[L]


def kl_two_gauss(mu0, sigma0, mu1, sigma1, level_weight, loss_weight):
    """level_weight * KL_two_gauss_with_diag_cov (phiseg_model.py:210-226, 271-279)."""
    return get_default_graph().add_op("kl", [mu0, sigma0, mu1, sigma1],
                                      dict(level_weight=float(level_weight), loss_weight=float(loss_weight)),
                                      [((), KIND_F32)], name="KL")[0]


def l2_of_collection(scale=1.0, name="weight_variables"):
    """scale * sum of tf.nn.l2_loss over a variable collection (phiseg_model.py:290-299, add_weight_decay)."""
    g = get_default_graph()
    return g.add_op("l2_weights", [], dict(vars=list(g.get_collection(name)), scale=float(scale)), [((), KIND_F32)],
                    name="weights_norm")[0]


def aggregate_logits(s_list):
    """_aggregate_output_list(use_softmax=False) + tf.nn.softmax (phiseg_model.py:106-109, 304-311)
    -> (sum of levels, softmax of the sum)."""
    outs = get_default_graph().add_op("aggregate", list(s_list), dict(L=len(s_list)),
                                      [(s_list[0].shape, KIND_F32), (s_list[0].shape, KIND_F32)], name="aggregate")
    return outs[0], outs[1]


def weighted_sum(scalars, weights):
    return get_default_graph().add_op("weighted_sum", list(scalars), dict(weights=[float(w) for w in weights]),
                                      [((), KIND_F32)], name="loss_tot")[0]
