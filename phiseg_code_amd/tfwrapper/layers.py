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
    if tuple(strides) != (1, 1) or padding != "SAME":
        raise NotImplementedError("hot path: stride 1, SAME padding")
    if tuple(kernel_size) not in ((1, 1), (3, 3)):
        raise NotImplementedError("hot path: 1x1 and 3x3 kernels")
    if normalise_post_activation or dropout_p is not None:
        raise NotImplementedError("normalise_post_activation / dropout are never set by the PHiSeg configs")
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
        head = kind is None and activation is not activations.relu
        return G.conv_unit(x, weights, biases, kernel_size[0], kind, norm_vars, activations.ACT_NAME[activation],
                           training, num_groups=kwargs.get('num_groups'), head=head, name='conv')


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
    """Channel concat of feature maps; the first defines the output size.  On the hot path all sizes are equal
    (prob_unet2D decoder, likelihoods.py:136), so no crop is ever needed."""
    out_size = inputs[0].get_shape().as_list()[1:3]
    for t in inputs[1:]:
        if t.get_shape().as_list()[1:3] != out_size:
            raise NotImplementedError("crop_and_concat with unequal sizes does not occur on the hot path")
    if axis not in (-1, 3):
        raise ValueError("crop_and_concat: channel axis only")
    out = inputs[0]
    for t in inputs[1:]:
        out = G.concat([out, t], axis=-1)
    return out


def _not_on_hot_path(name):
    def fn(*args, **kwargs):
        raise NotImplementedError("%s has zero call sites in phiseg/ (SURVEY.md section 2, out of scope)" % name)
    fn.__name__ = name
    return fn


maxpool2D = _not_on_hot_path("maxpool2D")
maxpool3D = _not_on_hot_path("maxpool3D")
reshape_pool2D_layer = _not_on_hot_path("reshape_pool2D_layer")
conv3D = _not_on_hot_path("conv3D")
transposed_conv3D = _not_on_hot_path("transposed_conv3D")
bilinear_upsample3D = _not_on_hot_path("bilinear_upsample3D")
dilated_conv2D = _not_on_hot_path("dilated_conv2D")
residual_unit2D = _not_on_hot_path("residual_unit2D")
identity_residual_unit2D = _not_on_hot_path("identity_residual_unit2D")
dense_layer = _not_on_hot_path("dense_layer")
pad_to_size = _not_on_hot_path("pad_to_size")
dropout = _not_on_hot_path("dropout")
