"""dev (GPU box): the XF rewrite on / off in one process under PHX_DETERMINISTIC=1: loss terms, gradients, parameters after a step."""
import os, sys
os.environ["PHX_DETERMINISTIC"] = "1"
sys.path.insert(0, ".")
import numpy as np, torch
from oracle import init as oinit, train as otrain
from phiseg_code_amd.phiseg import phiseg_model
from tests.helpers import load_golden
from tests.test_graph_cpu import make_config
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
g, cfg, var_order = load_golden("lidc_phiseg_bn")
cfg = dict(cfg, B=B)
params = otrain.make_params(var_order, cfg["weight_seed"], torch.float32, perturbed=True)
x_np, s_np = oinit.synthetic_batch(B, cfg["H"], cfg["nlabels"], cfg["data_seed"])
res = {}
for v in ("0", "1"):
    os.environ["PHX_XF"] = v
    model = phiseg_model.phiseg(make_config(cfg, "bf16"), rng_seed=cfg["eps_seed"])
    model.set_weights({k: t.detach().numpy() for k, t in params.items()})
    keys = sorted(model.loss_dict)
    plan = model.sess.plan_for([model.loss_dict[k] for k in keys], True, B, True) if False else None
    out = model.sess.run([model.train_step] + [model.loss_dict[k] for k in keys], {model.x_inp: x_np, model.s_inp: s_np, model.training_pl: True, model.lr_pl: lr})
    pl = list(model.sess.plans.values())[0]
    names = [getattr(fn, "__name__", "") for fn, _ in pl.launches]
    res[v] = (dict(zip(keys, [float(o) for o in out[1:]])), model.sess.store.export(grads=True), model.sess.store.export(),
              sum(n == "phx_conv3x3_mfma_bf16_xf" for n in names), sum("wgrad" in n and "xf" in n for n in names), len(names))
    del model, pl
l0, g0, p0, *c0 = res["0"]; l1, g1, p1, *c1 = res["1"]
print("launch counts", c0, c1)
for k in l0: print("%-36s %14.4f %14.4f  %s" % (k, l0[k], l1[k], "" if l0[k] == l1[k] else "DIFF"))
rows = []
for k in g0:
    d = np.abs(g0[k] - g1[k]).max(); n = np.abs(g0[k]).max()
    if d > 0: rows.append((d / max(n, 1e-30), k, d, n))
rows.sort(reverse=True)
print("gradients that differ: %d of %d" % (len(rows), len(g0)))
for r in rows[:25]: print("  %.3e  %-60s maxdiff %.3e max %.3e" % r)
rows = [(np.abs(p0[k] - p1[k]).max(), k) for k in p0 if np.abs(p0[k] - p1[k]).max() > 0]
rows.sort(reverse=True)
print("parameters / moving stats that differ: %d" % len(rows))
for r in rows[:10]: print("  %.3e  %s" % r)
