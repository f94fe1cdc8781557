"""Deterministic variable values, noise streams and synthetic LIDC-shaped inputs for the oracle
(test infrastructure only).  Everything derives from Philox streams so the build container (which
writes tests/golden/) and the GPU box (which re-creates the same inputs) agree without
committing weights.

Reference facts mirrored here:
* he_normal weights, zero biases: tfwrapper/utils.py:214-271 (R14);
* batch-norm variables gamma=1, beta=0, moving_mean=0, moving_variance=1 (tf.contrib.layers.batch_norm);
* group-norm gamma/beta 1/0, instance-norm scale~N(1,0.02)/offset 0: tfwrapper/normalisation.py:8-9,31-32;
* LIDC pixels are stored as image-0.5 and never re-normalised: data/lidc_data_loader.py:92,
  data/batch_provider.py:117-118; labels uint8 in [0, nlabels).
"""
import zlib

import numpy as np

from . import philox
from . import tf1_ops as T

NET_STREAM = {"posterior": 0, "prior": 1, "prior_gen": 2}


def stream_of(name):
    return zlib.crc32(name.encode()) & 0x3FFFFFFF


def variable_value(name, shape, seed=0, perturbed=True):
    """Value of TF variable `name` (see SURVEY.md Appendix B for the naming).  perturbed=True
    moves every affine parameter / statistic off its trivial init so parity tests exercise it."""
    st = stream_of(name)
    n = int(np.prod(shape))
    leaf = name.rsplit("/", 1)[-1]
    if leaf == "W":
        return T.he_normal_truncated(tuple(shape), seed, st)
    nrm = philox.normal(seed, 1, st, 1, n, dtype=np.float64)[0].reshape(shape)
    if not perturbed:
        if leaf in ("gamma", "moving_variance"):
            return np.ones(shape)
        if leaf == "scale":
            return 1.0 + 0.02 * nrm
        return np.zeros(shape)
    if leaf in ("gamma", "scale"):
        return 1.0 + 0.2 * nrm
    if leaf == "moving_variance":
        return 1.0 + 0.3 * philox.uniform01(seed, 2, st, n).reshape(shape)
    return 0.1 * nrm       # b, beta, offset, moving_mean


def eps_fn_numpy(seed, step, batch, sample_offset=0):
    """-> fn(net, level, shape) -> float64 ndarray; stream id = 16*net + level."""
    def fn(net, level, shape):
        per = int(np.prod(shape[1:]))
        e = philox.normal(seed, step, 16 * NET_STREAM[net] + level, shape[0], per,
                          sample_offset=sample_offset, dtype=np.float64)
        return e.reshape(shape)
    return fn


def synthetic_batch(batch, size, nlabels, seed=1234, step=0, sample_offset=0):
    """x [B,H,W,1] float32 ~ U(-0.5,0.5); s [B,H,W] uint8: nested filled ellipses (label k inside
    the k-th ellipse), 25 % empty masks -- lesion-like, SURVEY.md section 8(d)."""
    h = w = size
    xs, ss = [], []
    yy, xx = np.mgrid[0:h, 0:w]
    for b in range(batch):
        g = b + sample_offset
        u = philox.uniform01(seed, step, 1000 + g, h * w + 8)
        xs.append((u[: h * w] - 0.5).astype(np.float32).reshape(h, w, 1))
        p = u[h * w:]
        s = np.zeros((h, w), dtype=np.uint8)
        if p[0] >= 0.25:
            cy, cx = (0.3125 + 0.375 * p[1]) * h, (0.3125 + 0.375 * p[2]) * w
            ry, rx = (3 + 17 * p[3]) * h / 128.0, (3 + 17 * p[4]) * w / 128.0
            for k in range(1, nlabels):
                f = 1.0 - (k - 1) / float(nlabels - 1) if nlabels > 2 else 1.0
                inside = ((yy - cy) / (ry * f)) ** 2 + ((xx - cx) / (rx * f)) ** 2 <= 1.0
                s[inside] = k
        ss.append(s)
    return np.stack(xs), np.stack(ss)
