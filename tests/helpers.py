"""Shared test helpers: load golden fixtures, rebuild their inputs from the seeds (oracle side)."""
import json
import os

import numpy as np
import torch

from oracle import init as oinit
from oracle import train as otrain

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    alt = os.environ.get("PHX_TEST_GOLDEN_DIR")          # derived fixtures written by a test (e.g. a doubled batch)
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    if alt and os.path.exists(os.path.join(alt, name + ".npz")):
        path = os.path.join(alt, name + ".npz")
    g = np.load(path)
    cfg = json.loads(str(g["meta/cfg_json"]))
    cfg["image_size"] = (cfg["H"], cfg["H"], 1)
    var_order = [(n, tuple(s)) for n, s in json.loads(str(g["meta/var_order_json"]))]
    return g, cfg, var_order


def golden_inputs(cfg, var_order, dtype=torch.float64):
    params = otrain.make_params(var_order, cfg["weight_seed"], dtype, perturbed=True)
    x_np, s_np = oinit.synthetic_batch(cfg["B"], cfg["H"], cfg["nlabels"], cfg["data_seed"])
    return params, x_np, s_np


def check_tensor(g, key, arr, rtol, atol=0.0, what=""):
    """Compare `arr` with golden entry `key` (full tensor, or checksum + ::8 subsample)."""
    arr = np.asarray(arr, dtype=np.float64)
    if key in g.files:
        ref = g[key]
        np.testing.assert_allclose(arr, ref, rtol=rtol, atol=atol + rtol * np.abs(ref).max(), err_msg=what + key)
    else:
        sub = g[key + "@sub8"]
        np.testing.assert_allclose(arr[:, ::8, ::8, :], sub, rtol=rtol, atol=atol + rtol * np.abs(sub).max(),
                                   err_msg=what + key)
        np.testing.assert_allclose(np.abs(arr).sum(), float(g[key + "@abssum"]), rtol=max(rtol, 1e-9) * 10,
                                   err_msg=what + key + "@abssum")


# ---- validation metrics -----------------------------------------------------------------------
METRICS_CASES = [(11, 6, 4, 32, 32, 2, "plain"), (12, 16, 4, 48, 40, 4, "plain"), (13, 5, 3, 24, 24, 3, "empty_fg"),
                 (14, 4, 2, 16, 16, 2, "identical"), (15, 8, 1, 32, 32, 2, "plain")]


def metrics_case(seed, N, M, X, Y, C, mode):
    """Seeded inputs of the metrics fixtures (tools/make_goldens_metrics.py stores only the expected outputs): N blobby
    soft-max samples [N, X, Y, C] and M annotations [M, X, Y]; "empty_fg": some maps without foreground; "identical":
    all samples and annotations are the same map."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:X, 0:Y]

    def blob_logits(shift):
        lg = np.zeros((X, Y, C))
        for c in range(1, C):
            cx, cy, r = rng.uniform(0.25, 0.75) * X + shift, rng.uniform(0.25, 0.75) * Y, rng.uniform(0.1, 0.3) * X
            lg[..., c] = 3.0 - ((xx - cx) ** 2 + (yy - cy) ** 2) / (r * r) * 3.0
        return lg
    base = blob_logits(0.0)
    sm = np.zeros((N, X, Y, C))
    for i in range(N):
        lg = base + 0.8 * blob_logits(rng.normal() * 2.0) + rng.normal(size=(X, Y, C)) * 0.3
        if mode == "empty_fg" and i % 2 == 0:
            lg[..., 1:] -= 20.0
        e = np.exp(lg - lg.max(axis=-1, keepdims=True))
        sm[i] = e / e.sum(axis=-1, keepdims=True)
    gts = np.zeros((M, X, Y), dtype=np.uint8)
    for j in range(M):
        g = (base + 0.8 * blob_logits(rng.normal() * 2.0)).argmax(axis=-1)
        if mode == "empty_fg" and j == 0:
            g[:] = 0
        gts[j] = g
    if mode == "identical":
        sm[:] = sm[0]
        gts[:] = sm[0].argmax(axis=-1)
    return sm.astype(np.float32), gts
