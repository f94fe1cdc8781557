"""Diagnostic (GPU box): every conv_unit output of the training graph vs the oracle's intermediates."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from tests.test_model_gpu import build
from oracle import nets, train as otrain

case = sys.argv[1] if len(sys.argv) > 1 else "lidc_phiseg_bn"
dt = sys.argv[2] if len(sys.argv) > 2 else "f32"
g, cfg, var_order, model, params, x_np, s_np = build(case, dt)
rec = {}
orig = nets.Ctx.conv
def conv(self, x, scope, act="relu", normalise=True):
    y = orig(self, x, scope, act, normalise)
    rec.setdefault(scope, y.detach().numpy())
    return y
nets.Ctx.conv = conv
x = torch.as_tensor(x_np, dtype=torch.float64)
nets.elbo(params, x, torch.as_tensor(s_np), otrain.torch_eps_fn(cfg["eps_seed"], 0, cfg["B"]), cfg, training=True)
units = [op for op in model.graph.ops if op.type == "conv_unit" and op.name[:-5] in rec]
seen, fetch = set(), []
for op in units:
    if op.name not in seen:
        seen.add(op.name); fetch.append(op)
full = len(sys.argv) > 3
tensors = [op.outputs[0] for op in fetch] + (list(model.s_out_list) if full else [])
vals = model.sess.run(tensors, {model.x_inp: x_np, model.s_inp: s_np, model.training_pl: True})
for op, v in zip(fetch, vals):
    r = rec[op.name[:-5]]
    e = np.abs(v - r).max() / max(np.abs(r).max(), 1e-30)
    if e > 1e-4 or len(sys.argv) > 4:
        print("%-44s %-20s relerr %.3e" % (op.name, v.shape, e))
print("done", len(fetch))
