"""Diagnostic (GPU box): posterior input + first conv unit vs oracle at LIDC size."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from tests.test_model_gpu import build
from oracle import tf1_ops as T

g, cfg, var_order, model, params, x_np, s_np = build("lidc_phiseg_bn", "f32")
ops = {op.name: op for op in model.graph.ops}
names = [n for n in ops if n.startswith("posterior/") and ("z0_pre_1" in n or n.endswith("concat"))]
print(names[:10])
cat_t = [op for op in model.graph.ops if op.type == "concat"][0].outputs[0]
c1 = ops["posterior/z0_pre_1/conv"].outputs[0]
p1 = ops["prior/z0_pre_1/conv"].outputs[0]
cat, a1, b1 = model.sess.run([cat_t, c1, p1], {model.x_inp: x_np, model.s_inp: s_np, model.training_pl: True})
x = torch.as_tensor(x_np, dtype=torch.float64)
oh = T.one_hot(torch.as_tensor(s_np), 2, torch.float64)
ref_cat = torch.cat([x, oh - 0.5], dim=-1)
print("concat err", np.abs(cat - ref_cat.numpy()).max())
def unit(inp, pre):
    y = T.conv2d_same(inp, params[pre + "/W"].detach())
    bn = pre + "/batch_norm/BatchNorm/"
    y, m, v = T.batch_norm_train(y, params[bn + "gamma"].detach(), params[bn + "beta"].detach())
    return T.relu(y).numpy()
r = unit(ref_cat, "posterior/z0_pre_1")
print("posterior z0_pre_1 err", np.abs(a1 - r).max() / np.abs(r).max(), "label sum", s_np.sum())
r = unit(x, "prior/z0_pre_1")
print("prior z0_pre_1 err", np.abs(b1 - r).max() / np.abs(r).max())
