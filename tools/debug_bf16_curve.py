"""GPU box: per-term loss curves, bf16 HIP (free running, two runs) vs fp32 HIP, n0=32 128x128 B=2 -- where does the bf16 curve jump?"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import init as oinit, train as otrain
from tests.helpers import load_golden
from tests.test_graph_cpu import make_config
from phiseg_code_amd.phiseg import phiseg_model
g, cfg, var_order = load_golden("lidc_phiseg_bn")
params = otrain.make_params(var_order, cfg["weight_seed"], torch.float32, perturbed=True)
p0 = {k: v.detach().clone().numpy() for k, v in params.items()}
x, s = oinit.synthetic_batch(cfg["B"], cfg["H"], cfg["nlabels"], cfg["data_seed"])
def run(dt, n=8):
    m = phiseg_model.phiseg(make_config(cfg, dt), rng_seed=cfg["eps_seed"])
    m.set_weights(p0)
    keys = sorted(m.loss_dict)
    rows = []
    for _ in range(n):
        out = m.sess.run([m.train_step] + [m.loss_dict[k] for k in keys], {m.x_inp: x, m.s_inp: s, m.training_pl: True, m.lr_pl: 2e-5})
        rows.append([float(v) for v in out[1:]])
    return keys, np.array(rows)
keys, f = run("f32")
_, b1 = run("bf16")
_, b2 = run("bf16")
np.set_printoptions(linewidth=250, precision=3, suppress=True)
for i, k in enumerate(keys):
    print("%-34s f32 %s" % (k, f[:, i]))
    print("%-34s b/f %s" % ("", b1[:, i] / f[:, i]))
    print("%-34s b2/f %s" % ("", b2[:, i] / f[:, i]))
