"""Oracle training step: ELBO -> autograd gradients -> TF1-form Adam (test infrastructure only).

The reference obtains gradients and the update from ``optimizer.minimize(loss_tot)``
(phiseg/phiseg_model.py:135-141, tf.train.AdamOptimizer defaults); here torch-CPU autograd
differentiates the restated forward (oracle.nets.elbo) and ``tf1_ops.adam_tf1_step`` applies the
[TF1.12] epsilon-hat Adam.  Variables whose gradient is None (never-consumed branches)
are skipped exactly as TF skips them.
"""
import numpy as np
import torch

from . import init as oinit
from . import nets
from . import tf1_ops as T


def make_params(var_specs, seed=0, dtype=torch.float64, perturbed=True):
    """var_specs: iterable of (tf_name, shape).  -> dict name -> torch tensor (requires_grad for
    trainables, i.e. everything but moving statistics)."""
    p = {}
    for name, shape in var_specs:
        v = torch.as_tensor(oinit.variable_value(name, list(shape), seed, perturbed), dtype=dtype)
        v = v.reshape(list(shape)).clone()
        if not name.rsplit("/", 1)[-1].startswith("moving_"):
            v.requires_grad_(True)
        p[name] = v
    return p


def torch_eps_fn(seed, step, batch, dtype=torch.float64, sample_offset=0):
    f = oinit.eps_fn_numpy(seed, step, batch, sample_offset)
    return lambda net, level, shape: torch.as_tensor(f(net, level, shape), dtype=dtype)


def loss_and_grads(params, x, s, eps_fn, cfg):
    for v in params.values():
        if v.grad is not None:
            v.grad = None
    out = nets.elbo(params, x, s, eps_fn, cfg, training=True)
    out["loss_tot"].backward()
    grads = {k: v.grad for k, v in params.items() if v.requires_grad}
    return out, grads


def train_steps(params, batches, cfg, eps_seed, lr=1e-3, n_steps=1, dtype=torch.float64, snapshots=None):
    """Runs n_steps of (ELBO, backward, Adam, moving-stat update) in place; batches[i] = (x, s) numpy.
    Step index i is the Philox `step` word for the noise; Adam's t = i + 1.  Returns list of losses.
    snapshots: optional dict {step index: None}; filled with (params, adam m, adam v) copies taken BEFORE that step."""
    m = {k: torch.zeros_like(v) for k, v in params.items() if v.requires_grad}
    vv = {k: torch.zeros_like(v) for k, v in params.items() if v.requires_grad}
    losses = []
    for i in range(n_steps):
        if snapshots is not None and i in snapshots:
            snapshots[i] = ({k: v.detach().clone() for k, v in params.items()}, {k: v.clone() for k, v in m.items()},
                            {k: v.clone() for k, v in vv.items()})
        x_np, s_np = batches[i % len(batches)]
        x = torch.as_tensor(x_np, dtype=dtype)
        s = torch.as_tensor(s_np)
        out, grads = loss_and_grads(params, x, s, torch_eps_fn(eps_seed, i, x.shape[0], dtype), cfg)
        losses.append({k: float(v) for k, v in out["loss_dict"].items()})
        with torch.no_grad():
            for k, g in grads.items():
                if g is None:
                    continue
                pn, mn, vn = T.adam_tf1_step(params[k], g, m[k], vv[k], i + 1, lr)
                params[k].copy_(pn)
                m[k], vv[k] = mn, vn
            for k, nv in out["moving_updates"].items():
                params[k].copy_(nv)
    return losses
