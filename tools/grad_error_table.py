"""GPU box: per-variable gradient error of one bf16 training step of the engine against the oracle (exact fp32 and with the bf16
storage policy simulated) at a given batch -- the table behind tests/test_model_gpu.py::_bf16_plan_vs_oracle.
usage: python tools/grad_error_table.py [batch=64] [runs=2] [seed offset=0] [name filter]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from tests.helpers import load_golden
from tests.test_graph_cpu import make_config
from oracle import train as otrain, nets, init as oinit
from phiseg_code_amd.phiseg import phiseg_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
g, cfg, _ = load_golden("lidc_phiseg_bn")
off = int(sys.argv[3]) if len(sys.argv) > 3 else 0
flt = sys.argv[4] if len(sys.argv) > 4 else None
cfg = dict(cfg, B=B, weight_seed=cfg["weight_seed"] + off, eps_seed=cfg["eps_seed"] + off, data_seed=cfg["data_seed"] + off)
model = phiseg_model.phiseg(make_config(cfg, "bf16"), rng_seed=cfg["eps_seed"])
var_order = [(n, tuple(v.shape)) for n, v in model.graph.variables.items()]
params = otrain.make_params(var_order, cfg["weight_seed"], torch.float32, perturbed=True)
x_np, s_np = oinit.synthetic_batch(B, cfg["H"], cfg["nlabels"], cfg["data_seed"])
model.set_weights({k: v.detach().numpy() for k, v in params.items()})
xt, st = torch.as_tensor(x_np, dtype=torch.float32), torch.as_tensor(s_np)


def oracle_eval(sim):
    for v in params.values():
        v.grad = None
    eps = otrain.torch_eps_fn(cfg["eps_seed"], 0, B, torch.float32)
    out = nets.elbo(params, xt, st, eps, cfg, training=True, bf16_sim=sim)
    out["loss_tot"].backward()
    return ({k: float(v.detach()) for k, v in out["loss_dict"].items()},
            {k: v.grad.detach().double().numpy().copy() for k, v in params.items() if v.requires_grad and v.grad is not None})


t_exact, g_exact = oracle_eval(False)
t_sim, g_sim = oracle_eval(True)
keys = sorted(model.loss_dict)
plan = model.sess.plan_for([model.loss_dict[k] for k in keys], True, B, True)
plan.set_input("x_input", x_np)
plan.set_input("s_input", s_np)
model.sess.store.set_lr(0.0)
for r in range(runs):
    model.sess.store.step.zero_()                      # the Philox step word of a training plan: the same noise in every repetition
    plan.run()
    plan.sync()
    terms = {k: float(plan.fetch(model.loss_dict[k])) for k in keys}
    got = model.sess.store.export(grads=True)
    rows = []
    for name, ge in g_exact.items():
        nrm = np.linalg.norm(ge)
        if nrm < 1e-8 * max(1.0, np.sqrt(ge.size)):
            continue
        gh = got[name].astype(np.float64).reshape(ge.shape)
        inh = np.linalg.norm(g_sim[name] - ge) / nrm
        e = np.linalg.norm(gh - ge) / nrm
        es = np.linalg.norm(gh - g_sim[name]) / nrm
        cos = float((gh * ge).sum() / (np.linalg.norm(gh) * nrm))
        rows.append((e / max(inh, 0.03), name, ge.size, e, inh, es, np.linalg.norm(gh) / nrm, cos))
    rows.sort(reverse=True)
    print("run %d: terms (engine / exact / simulated):" % r, {k: (round(terms[k], 3), round(t_exact[k], 3), round(t_sim[k], 3)) for k in keys})
    print("mean e %.4f mean inh %.4f" % (np.mean([x[3] for x in rows]), np.mean([x[4] for x in rows])))
    for x in (rows[:10] if flt is None else [r_ for r_ in rows if flt in r_[1]]):
        print("  %5.2fx  %-46s n=%-7d e %.4f  sim %.4f  e_vs_sim %.4f  |g|/|g_exact| %.4f  cos %.5f" % x)
