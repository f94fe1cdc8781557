"""Diagnostic (GPU box): per-tensor relative errors of the HIP engine vs the golden fixtures."""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.test_model_gpu import build

case, dt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "f32")
g, cfg, var_order, model, params, x_np, s_np = build(case, dt)
L = cfg["latent_levels"]
fetch = [model.mu_list, model.sigma_list, model.z_list, model.prior_mu_list, model.prior_sigma_list, model.s_out_list]
names = ["mu", "sigma", "z", "prior_mu", "prior_sigma", "s"]
out = model.sess.run(fetch, {model.x_inp: x_np, model.s_inp: s_np, model.training_pl: True})
for nm, lst in zip(names, out):
    for l in reversed(range(L)):
        key = "train/%s_%d" % (nm, l)
        a = np.asarray(lst[l], dtype=np.float64)
        if key in g.files:
            ref = g[key]
        else:
            ref, a = g[key + "@sub8"], a[:, ::8, ::8, :]
        print("%-14s l=%d shape=%-18s relerr=%.3e" % (nm, l, a.shape, np.abs(a - ref).max() / np.abs(ref).max()))
