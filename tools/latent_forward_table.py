"""GPU box: forward latent heads of one bf16 training-mode pass at a given batch -- engine against the oracle (exact fp32 and with the bf16
storage policy simulated): per level and channel the pixel means of mu_q, mu_p, sigma_p, (mu_q - mu_p) / sigma_p^2 (what the KL
gradient of the prior's mu head sums) and the relative L2 distance of the maps.  usage: python tools/latent_forward_table.py [batch=64] [seed offset=0]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from tests.helpers import load_golden
from tests.test_graph_cpu import make_config
from oracle import train as otrain, nets, init as oinit
from phiseg_code_amd.phiseg import phiseg_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
off = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g, cfg, _ = load_golden("lidc_phiseg_bn")
cfg = dict(cfg, B=B, weight_seed=cfg["weight_seed"] + off, eps_seed=cfg["eps_seed"] + off, data_seed=cfg["data_seed"] + off)
model = phiseg_model.phiseg(make_config(cfg, "bf16"), rng_seed=cfg["eps_seed"])
var_order = [(n, tuple(v.shape)) for n, v in model.graph.variables.items()]
params = otrain.make_params(var_order, cfg["weight_seed"], torch.float32, perturbed=True)
x_np, s_np = oinit.synthetic_batch(B, cfg["H"], cfg["nlabels"], cfg["data_seed"])
model.set_weights({k: v.detach().numpy() for k, v in params.items()})
xt, st = torch.as_tensor(x_np, dtype=torch.float32), torch.as_tensor(s_np)
L = cfg["latent_levels"]


def oracle(sim):
    with torch.no_grad():
        eps = otrain.torch_eps_fn(cfg["eps_seed"], 0, B, torch.float32)
        out = nets.elbo(params, xt, st, eps, cfg, training=True, bf16_sim=sim)
    return {k: [t.double().numpy() for t in out[k]] for k in ("mu", "sigma", "prior_mu", "prior_sigma")}


ex, sm = oracle(False), oracle(True)
fetch = list(model.mu_list) + list(model.sigma_list) + list(model.prior_mu_list) + list(model.prior_sigma_list)
plan = model.sess.plan_for(fetch + [model.loss_tot], True, B, True)
plan.set_input("x_input", x_np)
plan.set_input("s_input", s_np)
model.sess.store.set_lr(0.0)
model.sess.store.step.zero_()
plan.run()
plan.sync()
vals = [np.asarray(plan.fetch(t), dtype=np.float64) for t in fetch]
en = dict(mu=vals[0:L], sigma=vals[L:2 * L], prior_mu=vals[2 * L:3 * L], prior_sigma=vals[3 * L:4 * L])


def stats(d, l):
    dl = d["mu"][l] - d["prior_mu"][l]
    gsum = (dl / d["prior_sigma"][l] ** 2)
    return dict(mu_q=d["mu"][l].mean((0, 1, 2)), mu_p=d["prior_mu"][l].mean((0, 1, 2)), sig_p=d["prior_sigma"][l].mean((0, 1, 2)),
                delta=dl.mean((0, 1, 2)), g=gsum.mean((0, 1, 2)), g_abs=np.abs(gsum).mean((0, 1, 2)))


for l in range(L):
    print("level %d  map %s" % (l, ex["mu"][l].shape))
    for name, d in (("exact", ex), ("simulated", sm), ("engine", en)):
        s_ = stats(d, l)
        print("  %-9s mean mu_q %s  mu_p %s  sigma_p %s  delta %s  delta/sigma_p^2 %s  (mean |.| %s)" %
              (name, np.round(s_["mu_q"], 5), np.round(s_["mu_p"], 5), np.round(s_["sig_p"], 5), np.round(s_["delta"], 5), np.round(s_["g"], 5), np.round(s_["g_abs"], 4)))
    for k in ("mu", "prior_mu", "sigma", "prior_sigma"):
        n = np.linalg.norm(ex[k][l])
        print("  %-12s rel. L2: simulated - exact %.4f   engine - exact %.4f   engine - simulated %.4f" %
              (k, np.linalg.norm(sm[k][l] - ex[k][l]) / n, np.linalg.norm(en[k][l] - ex[k][l]) / n, np.linalg.norm(en[k][l] - sm[k][l]) / n))
    for name, d in (("simulated", sm), ("engine", en)):
        de, dd = ex["mu"][l] - ex["prior_mu"][l], d["mu"][l] - d["prior_mu"][l]
        print("  delta map %-9s: rel. L2 to exact %.4f   <d, d_exact> / <d_exact, d_exact> %.4f" % (name, np.linalg.norm(dd - de) / np.linalg.norm(de), float((dd * de).sum() / (de * de).sum())))
