"""Variable factories -- counterpart of the reference's tfwrapper/utils.py:214-271.

``get_weight_variable`` / ``get_bias_variable`` keep the reference's signatures.  he_normal follows TF 1.12's
``variance_scaling_initializer(factor=2.0, mode='FAN_IN', uniform=False)``: a truncated normal (resampled
beyond two standard deviations) with stddev sqrt(1.3 * 2 / fan_in), fan_in = kh*kw*Cin."""
import math

import numpy as np

from phiseg_code_amd import graph as G


def _truncated_normal(shape, std, rng):
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return out * std


def _fans(shape):
    rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    return rf * shape[-2], rf * shape[-1]


def _initializer(type_, **kwargs):
    if type_ == "he_normal":
        return lambda shape, rng: _truncated_normal(shape, math.sqrt(1.3 * 2.0 / _fans(shape)[0]), rng)
    if type_ == "he_uniform":
        return lambda shape, rng: rng.uniform(-1, 1, shape) * math.sqrt(3.0 * 2.0 / _fans(shape)[0])
    if type_ == "caffe_uniform":
        return lambda shape, rng: rng.uniform(-1, 1, shape) * math.sqrt(3.0 * 1.0 / _fans(shape)[0])
    if type_ == "xavier_uniform":
        return lambda shape, rng: rng.uniform(-1, 1, shape) * math.sqrt(6.0 / sum(_fans(shape)))
    if type_ == "xavier_normal":
        return lambda shape, rng: _truncated_normal(shape, math.sqrt(1.3 * 2.0 / sum(_fans(shape))), rng)
    if type_ == "simple":
        std = kwargs.get("stddev", 0.02)
        return lambda shape, rng: _truncated_normal(shape, std, rng)
    raise ValueError("Unknown initialisation requested: %s" % type_)


def get_weight_variable(shape, name=None, type="xavier_uniform", regularize=True, **kwargs):
    if name is None:
        raise NotImplementedError("unnamed variables are not used on the hot path")
    if kwargs.get("init_weights") is not None:
        w0 = np.asarray(kwargs["init_weights"], dtype=np.float32)
        init = lambda shape, rng: w0
    else:
        init = _initializer(type, **kwargs)
    g = G.get_default_graph()
    weight = g.get_variable(name, shape, init)
    if regularize:
        g.add_to_collection("weight_variables", weight)
    return weight


def get_bias_variable(shape, name=None, init_value=0.0, **kwargs):
    if name is None:
        raise NotImplementedError("unnamed variables are not used on the hot path")
    if kwargs.get("init_biases") is not None:
        b0 = np.asarray(kwargs["init_biases"], dtype=np.float32)
        init = lambda shape, rng: b0
    else:
        init = lambda shape, rng: np.full(shape, init_value, dtype=np.float32)
    return G.get_default_graph().get_variable(name, shape, init)


def get_rhs_dim(tensor):
    return int(np.prod(tensor.get_shape().as_list()[1:]))


def flatten(tensor):
    raise NotImplementedError("flatten is folded into the KL kernel (phx_kl_diag_gauss)")
