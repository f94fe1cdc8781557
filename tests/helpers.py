"""Shared test helpers: load golden fixtures, rebuild their inputs from the seeds (oracle side)."""
import json
import os

import numpy as np
import torch

from oracle import init as oinit
from oracle import train as otrain

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
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
