"""Normalisation selectors -- counterpart of the reference's tfwrapper/normalisation.py.

The functions keep the reference's names and signatures.  ``layers.conv2D(normalisation=...)`` recognises
them by identity (exactly like the reference's ``normalisation is tfnorm.batch_norm`` test,
tfwrapper/layers.py:126) and fuses conv -> norm -> activation into one ``conv_unit`` node; ``make_variables``
creates the variables under the same scope names TF would."""
import numpy as np

from phiseg_code_amd import graph as G

_ones = lambda shape, rng: np.ones(shape, dtype=np.float32)
_zeros = lambda shape, rng: np.zeros(shape, dtype=np.float32)


def _standalone(kind):
    def fn(x, **kwargs):
        raise NotImplementedError("%s is applied through layers.conv2D(normalisation=...) on the hot path" % kind)
    return fn


def batch_norm(x, training=None, moving_average_decay=0.99, scope="batch_norm", **kwargs):
    """tf.contrib.layers.batch_norm(decay=.99, epsilon=1e-3, center, scale) -- normalisation.py:145-163."""
    return _standalone("batch_norm")(x)


def group_norm2D(x, eps=1e-5, scope="group_norm", **kwargs):
    """normalisation.py:17-36: G = kwargs['num_groups'] or max(2, C // 16)."""
    return _standalone("group_norm2D")(x)


def instance_norm2D(x, scope="instance_norm", **kwargs):
    """normalisation.py:3-14."""
    return _standalone("instance_norm2D")(x)


def identity(x, **kwargs):
    """normalisation.py:166-171."""
    return x


def layer_norm(x, **kwargs):
    raise NotImplementedError("layer_norm has no call site in phiseg/ (out of scope, SURVEY.md section 2)")


def batch_renorm(x, **kwargs):
    raise NotImplementedError("batch_renorm has no call site in phiseg/ (out of scope, SURVEY.md section 2)")


KIND = {batch_norm: "batch", group_norm2D: "group", instance_norm2D: "instance", identity: None}
EPS = {"batch": 1e-3, "group": 1e-5, "instance": 1e-5}
BN_DECAY = 0.99


def make_variables(kind, channels, scope=None):
    """Create the norm's variables inside the CURRENT scope (the conv's name scope); `scope` replaces the callable's default
    scope name (batch_norm / group_norm / instance_norm), as the residual units pass scope='bn1' (layers.py:452-456)."""
    g = G.get_default_graph()
    if kind == "batch":
        with g.variable_scope(scope or "batch_norm"):
            with g.variable_scope("BatchNorm"):
                return dict(beta=g.get_variable("beta", [channels], _zeros),
                            gamma=g.get_variable("gamma", [channels], _ones),
                            moving_mean=g.get_variable("moving_mean", [channels], _zeros, trainable=False),
                            moving_variance=g.get_variable("moving_variance", [channels], _ones, trainable=False))
    if kind == "group":
        with g.variable_scope(scope or "group_norm"):
            return dict(gamma=g.get_variable("gamma", [1, 1, 1, channels], _ones),
                        beta=g.get_variable("beta", [1, 1, 1, channels], _zeros))
    if kind == "instance":
        with g.variable_scope(scope or "instance_norm"):
            return dict(gamma=g.get_variable("scale", [channels], lambda s, rng: 1.0 + 0.02 * rng.standard_normal(s)),
                        beta=g.get_variable("offset", [channels], _zeros))
    return {}
