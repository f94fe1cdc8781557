"""Layer wrappers -- counterpart of the reference's tfwrapper/layers.py with the same signatures.

Hot-path layers (surface B1): ``conv2D`` (layers.py:94-145), ``averagepool2D`` (44-54),
``bilinear_upsample2D`` (336-345), ``global_averagepool2D`` (70-78), ``crop_and_concat`` (586-622),
``nearest_neighbour_upsample2D`` (326-333).  Every call adds nodes to ``phiseg_code_amd.graph``; the arithmetic
runs in hand-written HIP kernels (libphx.so).  Layers that no PHiSeg configuration calls (3-D variants,
residual units, dense, dilated conv) keep their names and raise NotImplementedError; transposed_conv2D (named by the
north star) is implemented on direct kernels.
"""
import logging

import numpy as np

from phiseg_code_amd import graph as G
from phiseg_code_amd.tfwrapper import activations
from phiseg_code_amd.tfwrapper import normalisation as tfnorm
from phiseg_code_amd.tfwrapper import utils

# Will be used as default in all the layers below (reference: tf.nn.relu)
STANDARD_NONLINEARITY = activations.relu


def averagepool2D(x, kernel_size=(2, 2), strides=(2, 2), padding="SAME"):
    """tf.nn.avg_pool 2x2 / stride 2 / SAME (odd sizes: divide by the number of valid taps)."""
    if tuple(kernel_size) != (2, 2) or tuple(strides) != (2, 2) or padding != "SAME":
        raise NotImplementedError("only the 2x2 / stride 2 / SAME average pool is on the hot path")
    return G.avg_pool2x2(x)


def global_averagepool2D(x, name=None):
    return G.global_average_pool(x, name=name)


def conv2D(x,
           name,
           kernel_size=(3, 3),
           num_filters=32,
           strides=(1, 1),
           activation=STANDARD_NONLINEARITY,
           normalisation=tfnorm.identity,
           normalise_post_activation=False,
           dropout_p=None,
           padding="SAME",
           weight_init='he_normal',
           add_bias=True,
           **kwargs):
    """Standard 2-D convolutional layer: conv -> [bias] -> normalisation -> activation.
    kwargs can carry ``training`` and normalisation parameters (``num_groups``)."""
    if padding != "SAME":
        raise NotImplementedError("conv2D: SAME padding (the only value the reference's call sites use)")
    if normalise_post_activation:
        raise NotImplementedError("normalise_post_activation is never set by the PHiSeg configs")
    # stride 1 with a 1x1 / 3x3 kernel is the hot path (MFMA / direct / head kernels); anything else -- strided convolutions of the
    # residual units, larger kernels -- runs on the general direct kernels of csrc/gconv.hip
    general = None
    if tuple(strides) != (1, 1) or tuple(kernel_size) not in ((1, 1), (3, 3)):
        general = (int(kernel_size[0]), int(kernel_size[1]), int(strides[0]), int(strides[1]), 1, 1)
    if normalisation not in tfnorm.KIND:
        raise ValueError("Unknown normalisation callable %r" % (normalisation,))
    if activation not in activations.ACT_NAME:
        raise ValueError("Unknown activation callable %r" % (activation,))

    bottom_num_filters = x.get_shape().as_list()[-1]
    weight_shape = [kernel_size[0], kernel_size[1], bottom_num_filters, num_filters]
    bias_shape = [num_filters]
    g = G.get_default_graph()

    with g.variable_scope(name):
        weights = utils.get_weight_variable(weight_shape, name='W', type=weight_init, regularize=True)
        biases = None
        if add_bias and normalisation is tfnorm.batch_norm:
            logging.debug('Turning of bias because using batch norm.')
            add_bias = False
        if add_bias:
            biases = utils.get_bias_variable(bias_shape, name='b')
        kind = tfnorm.KIND[normalisation]
        norm_vars = tfnorm.make_variables(kind, num_filters)
        training = kwargs.get('training', True)
        # heads (mu / sigma / logits) keep fp32 storage in the bf16 configuration
        head = kind is None and activation is not activations.relu and general is None
        op = G.conv_unit(x, weights, biases, kernel_size[0], kind, norm_vars, activations.ACT_NAME[activation],
                         training, num_groups=kwargs.get('num_groups'), head=head, name='conv', general=general)
        if dropout_p is not None:
            op = dropout(op, keep_prob=dropout_p, training=training)
        return op


def dilated_conv2D(bottom,
                   name,
                   kernel_size=(3, 3),
                   num_filters=32,
                   rate=2,
                   activation=STANDARD_NONLINEARITY,
                   normalisation=tfnorm.identity,
                   normalise_post_activation=False,
                   dropout_p=None,
                   padding="SAME",
                   weight_init='he_normal',
                   add_bias=True,
                   **kwargs):
    """tfwrapper/layers.py:378-425: tf.nn.atrous_conv2d(bottom, W, rate, SAME) -> [bias] -> normalisation -> activation
    [-> dropout].  (Unlike conv2D the bias is kept in front of batch_norm, layers.py:406-409.)"""
    if padding != "SAME" or normalise_post_activation:
        raise NotImplementedError("dilated_conv2D: SAME padding, normalisation before the activation")
    if normalisation not in tfnorm.KIND or activation not in activations.ACT_NAME:
        raise ValueError("Unknown normalisation / activation callable")
    cin = bottom.get_shape().as_list()[3]
    g = G.get_default_graph()
    with g.variable_scope(name):
        weights = utils.get_weight_variable([kernel_size[0], kernel_size[1], cin, num_filters], name='W', type=weight_init,
                                            regularize=True)
        biases = utils.get_bias_variable([num_filters], name='b') if add_bias else None
        kind = tfnorm.KIND[normalisation]
        norm_vars = tfnorm.make_variables(kind, num_filters)
        training = kwargs.get('training', True)
        op = G.conv_unit(bottom, weights, biases, kernel_size[0], kind, norm_vars, activations.ACT_NAME[activation], training,
                         num_groups=kwargs.get('num_groups'), head=False, name='atrous',
                         general=(int(kernel_size[0]), int(kernel_size[1]), 1, 1, int(rate), int(rate)))
        if dropout_p is not None:
            op = dropout(op, keep_prob=dropout_p, training=training)
        return op


def dense_layer(bottom,
                name,
                hidden_units=512,
                activation=STANDARD_NONLINEARITY,
                normalisation=tfnorm.batch_norm,
                normalise_post_activation=False,
                dropout_p=None,
                weight_init='he_normal',
                add_bias=True,
                **kwargs):
    """tfwrapper/layers.py:539-582: flatten -> matmul with W [F, hidden_units] -> [bias] -> normalisation -> activation
    [-> dropout].  Runs as a 1x1 convolution of the flattened input; the result keeps two unit axes: [B, 1, 1, hidden_units]
    (same memory as the reference's [B, hidden_units])."""
    if normalise_post_activation:
        raise NotImplementedError("dense_layer: normalisation before the activation only")
    if normalisation not in tfnorm.KIND or activation not in activations.ACT_NAME:
        raise ValueError("Unknown normalisation / activation callable")
    flat = G.flatten(bottom)
    f = flat.get_shape().as_list()[3]
    g = G.get_default_graph()
    with g.variable_scope(name):
        weights = utils.get_weight_variable([f, hidden_units], name='W', type=weight_init, regularize=True)
        biases = utils.get_bias_variable([hidden_units], name='b') if add_bias else None
        kind = tfnorm.KIND[normalisation]
        norm_vars = tfnorm.make_variables(kind, hidden_units)
        training = kwargs.get('training', True)
        op = G.conv_unit(flat, weights, biases, 1, kind, norm_vars, activations.ACT_NAME[activation], training,
                         num_groups=kwargs.get('num_groups'), head=False, name='dense', general=(1, 1, 1, 1, 1, 1))
        if dropout_p is not None:
            op = dropout(op, keep_prob=dropout_p, training=training)
        return op


def _conv_norm(x, conv_name, norm_scope, num_filters, strides, kernel_size, normalisation, activation, add_bias, kwargs):
    """conv2D(x, conv_name, activation=identity, add_bias=add_bias) -> normalisation(., scope=norm_scope) -> activation, as the
    residual units spell it (layers.py:452-460): the convolution's variables live under conv_name, the normalisation's under
    norm_scope, and the bias is kept even in front of batch norm."""
    g = G.get_default_graph()
    cin = x.get_shape().as_list()[-1]
    with g.variable_scope(conv_name):
        weights = utils.get_weight_variable([kernel_size[0], kernel_size[1], cin, num_filters], name='W', type='he_normal',
                                            regularize=True)
        biases = utils.get_bias_variable([num_filters], name='b') if add_bias else None
    kind = tfnorm.KIND[normalisation]
    norm_vars = tfnorm.make_variables(kind, num_filters, scope=norm_scope)
    general = None
    if tuple(strides) != (1, 1) or tuple(kernel_size) not in ((1, 1), (3, 3)):
        general = (int(kernel_size[0]), int(kernel_size[1]), int(strides[0]), int(strides[1]), 1, 1)
    return G.conv_unit(x, weights, biases, kernel_size[0], kind, norm_vars, activations.ACT_NAME[activation],
                       kwargs.get('training', True), num_groups=kwargs.get('num_groups'), head=False, name=conv_name, general=general)


def _residual_skip(x, num_filters, down_sample, projection, strides, activation, normalisation, add_bias, kwargs):
    """the skip path both residual units share (layers.py:462-472, 520-530)"""
    cin = x.get_shape().as_list()[-1]
    if cin == num_filters and not down_sample:
        return x
    if projection:
        return _conv_norm(x, 'projection', 'bn_projection', num_filters, strides, (1, 1), normalisation, activation, add_bias, kwargs)
    pad = (num_filters - cin) // 2
    _, h, w, _ = x.get_shape().as_list()
    s = 2 if down_sample else 1
    # tf.pad along the channel axis, then identity[:, ::2, ::2, :]: one strided window
    return G.window4(x, -(-h // s), -(-w // s), cin + 2 * pad, stride=(s, s), off=(0, 0, -pad), name='identity')


def residual_unit2D(x,
                    name,
                    num_filters=32,
                    down_sample=False,
                    projection=False,
                    activation=STANDARD_NONLINEARITY,
                    normalisation=tfnorm.batch_norm,
                    add_bias=True,
                    **kwargs):
    """tfwrapper/layers.py:428-478 (https://arxiv.org/abs/1512.03385): conv1 -> bn1 -> act -> conv2 -> bn2, + skip, -> act."""
    if normalisation not in tfnorm.KIND or activation not in activations.ACT_NAME:
        raise ValueError("Unknown normalisation / activation callable")
    strides = (2, 2) if down_sample else (1, 1)
    g = G.get_default_graph()
    with g.variable_scope(name):
        conv1 = _conv_norm(x, 'conv1', 'bn1', num_filters, strides, (3, 3), normalisation, activation, add_bias, kwargs)
        conv2 = _conv_norm(conv1, 'conv2', 'bn2', num_filters, (1, 1), (3, 3), normalisation, activations.identity, add_bias, kwargs)
        skip = _residual_skip(x, num_filters, down_sample, projection, strides, activation, normalisation, add_bias, kwargs)
        if skip.get_shape().as_list()[-1] != num_filters:
            raise ValueError("residual_unit2D: channel padding needs an even difference (%d -> %d)"
                             % (x.get_shape().as_list()[-1], num_filters))
        return G.add_act(skip, conv2, activations.ACT_NAME[activation], name='add')


def identity_residual_unit2D(x,
                             name,
                             num_filters,
                             down_sample=False,
                             projection=True,
                             activation=STANDARD_NONLINEARITY,
                             normalisation=tfnorm.batch_norm,
                             add_bias=True,
                             **kwargs):
    """tfwrapper/layers.py:481-536 (identity mappings, pre-activation order): bn1 -> act -> conv1 -> bn2 -> act -> conv2, + skip."""
    if normalisation not in tfnorm.KIND or activation not in activations.ACT_NAME:
        raise ValueError("Unknown normalisation / activation callable")
    cin = x.get_shape().as_list()[-1]
    if not projection:
        assert (cin == num_filters) or (cin * 2 == num_filters), \
            'Number of filters must remain constant, or be increased by a ' \
            'factor of 2. In filters: %d, Out filters: %d' % (cin, num_filters)
    strides = (2, 2) if down_sample else (1, 1)
    training = kwargs.get('training', True)
    kind = tfnorm.KIND[normalisation]
    an = activations.ACT_NAME[activation]
    g = G.get_default_graph()

    def conv(t, cname, st):
        with g.variable_scope(cname):
            ci = t.get_shape().as_list()[-1]
            weights = utils.get_weight_variable([3, 3, ci, num_filters], name='W', type='he_normal', regularize=True)
            biases = utils.get_bias_variable([num_filters], name='b') if add_bias else None
        general = (3, 3, st[0], st[1], 1, 1) if tuple(st) != (1, 1) else None
        return G.conv_unit(t, weights, biases, 3, None, {}, 'identity', training, head=False, name=cname, general=general)
    with g.variable_scope(name):
        op1 = G.norm_act(x, kind, tfnorm.make_variables(kind, cin, scope='bn1'), an, training, kwargs.get('num_groups'), name='bn1')
        op1 = conv(op1, 'conv1', strides)
        op2 = G.norm_act(op1, kind, tfnorm.make_variables(kind, num_filters, scope='bn2'), an, training, kwargs.get('num_groups'),
                         name='bn2')
        op2 = conv(op2, 'conv2', (1, 1))
        skip = _residual_skip(x, num_filters, down_sample, projection, strides, activation, normalisation, add_bias, kwargs)
        return G.add_act(skip, op2, 'identity', name='add')


def reshape_pool2D_layer(x):
    """tfwrapper/layers.py:57-67: space-to-depth by strided slices, concat([x[:,0::2,0::2], x[:,1::2,0::2], x[:,0::2,1::2],
    x[:,1::2,1::2]], axis=3)."""
    _, h, w, c = x.get_shape().as_list()
    if h % 2 or w % 2:
        raise ValueError("reshape_pool2D_layer: even spatial sizes (tf.concat of unequal slices fails in the reference too)")
    parts = [G.window4(x, h // 2, w // 2, c, stride=(2, 2), off=(oy, ox, 0), name='slice') for (oy, ox) in ((0, 0), (1, 0), (0, 1), (1, 1))]
    out = parts[0]
    for t in parts[1:]:
        out = G.concat([out, t], axis=-1)
    return out


def maxpool2D(x, kernel_size=(2, 2), strides=(2, 2), padding="SAME"):
    """tf.nn.max_pool 2x2 / stride 2 / SAME (tfwrapper/layers.py:18-28)."""
    if tuple(kernel_size) != (2, 2) or tuple(strides) != (2, 2) or padding != "SAME":
        raise NotImplementedError("maxpool2D: 2x2 / stride 2 / SAME (the reference's defaults)")
    return G.max_pool2x2(x)


def pad_to_size(bottom, output_size):
    """tfwrapper/layers.py:625-650: zero-pad the spatial axes of `bottom` to output_size = [B, H, W, C] (the odd pixel goes to the
    bottom / right)."""
    shp = bottom.get_shape().as_list()
    if len(shp) != 4:
        raise NotImplementedError('pad_to_size has not been extended to 3D (neither in the reference)')
    dy, dx = int(output_size[1]) - shp[1], int(output_size[2]) - shp[2]
    if dy < 0 or dx < 0:
        raise ValueError("pad_to_size: output smaller than input")
    return G.spatial_window(bottom, output_size[1], output_size[2], -(dy // 2), -(dx // 2), name='pad_to_size')


def dropout(bottom, keep_prob, training):
    """tfwrapper/layers.py:653-668: tf.nn.dropout(bottom, keep_prob) while training, identity otherwise."""
    g = G.get_default_graph()
    with g.variable_scope('dropout_layer'):
        return G.dropout(bottom, keep_prob, training)


def transposed_conv2D(bottom,
                      name,
                      kernel_size=(4, 4),
                      num_filters=32,
                      strides=(2, 2),
                      output_shape=None,
                      activation=STANDARD_NONLINEARITY,
                      normalisation=tfnorm.identity,
                      normalise_post_activation=False,
                      dropout_p=None,
                      padding="SAME",
                      weight_init='he_normal',
                      add_bias=True,
                      **kwargs):
    """Standard 2-D transposed convolution (tfwrapper/layers.py:197-258): tf.nn.conv2d_transpose with a [kh, kw, num_filters,
    bottom channels] filter, default behaviour up-samples by a factor of 2; -> [bias] -> normalisation -> activation.  Unlike
    conv2D the reference keeps the bias here even in front of batch_norm (layers.py:231-234)."""
    if padding != "SAME":
        raise NotImplementedError("transposed_conv2D: SAME padding (the reference's default) only")
    if normalise_post_activation or dropout_p is not None:
        raise NotImplementedError("normalise_post_activation / dropout are never set by the PHiSeg configs")
    if normalisation not in tfnorm.KIND:
        raise ValueError("Unknown normalisation callable %r" % (normalisation,))
    if activation not in activations.ACT_NAME:
        raise ValueError("Unknown activation callable %r" % (activation,))
    shp = bottom.get_shape().as_list()
    if output_shape is not None and tuple(output_shape[1:3]) != (shp[1] * strides[0], shp[2] * strides[1]):
        raise NotImplementedError("transposed_conv2D: output_shape other than input * strides")
    weight_shape = [kernel_size[0], kernel_size[1], num_filters, shp[3]]
    g = G.get_default_graph()
    with g.variable_scope(name):
        weights = utils.get_weight_variable(weight_shape, name='W', type=weight_init, regularize=True)
        biases = utils.get_bias_variable([num_filters], name='b') if add_bias else None
        kind = tfnorm.KIND[normalisation]
        norm_vars = tfnorm.make_variables(kind, num_filters)
        return G.conv_unit(bottom, weights, biases, kernel_size[0], kind, norm_vars, activations.ACT_NAME[activation],
                           kwargs.get('training', True), num_groups=kwargs.get('num_groups'), head=False, name='deconv',
                           transposed=(int(kernel_size[0]), int(kernel_size[1]), int(strides[0]), int(strides[1])))


def nearest_neighbour_upsample2D(x, factor):
    shp = x.get_shape().as_list()
    return G.resize_nearest(x, (shp[1] * factor, shp[2] * factor))


def bilinear_upsample2D(x, name, factor):
    """tf.image.resize_images(x, [f*h, f*w]) = TF 1.12 ResizeBilinear(align_corners=False), legacy coordinates."""
    if factor != 2:
        raise NotImplementedError("hot path: factor 2")
    with G.get_default_graph().variable_scope(name):
        return G.bilinear_up2x(x, name="ResizeBilinear")


def crop_and_concat(inputs, axis=-1):
    """tfwrapper/layers.py:586-622: the first feature map defines the output size, the others are centre-cropped to it
    (start = (larger - output) // 2) and everything is concatenated along the channel axis."""
    if axis not in (-1, 3):
        raise ValueError("crop_and_concat: channel axis only")
    out_size = inputs[0].get_shape().as_list()[1:3]
    out = inputs[0]
    for t in inputs[1:]:
        larger = t.get_shape().as_list()[1:3]
        if larger != out_size:
            if larger[0] < out_size[0] or larger[1] < out_size[1]:
                raise ValueError("crop_and_concat: %s cannot be cropped to %s" % (larger, out_size))
            t = G.spatial_window(t, out_size[0], out_size[1], (larger[0] - out_size[0]) // 2, (larger[1] - out_size[1]) // 2,
                                 name='crop')
        out = G.concat([out, t], axis=-1)
    return out


def _not_on_hot_path(name):
    def fn(*args, **kwargs):
        raise NotImplementedError("%s has zero call sites in phiseg/ (SURVEY.md section 2, out of scope)" % name)
    fn.__name__ = name
    return fn


maxpool3D = _not_on_hot_path("maxpool3D")
conv3D = _not_on_hot_path("conv3D")
transposed_conv3D = _not_on_hot_path("transposed_conv3D")
bilinear_upsample3D = _not_on_hot_path("bilinear_upsample3D")
