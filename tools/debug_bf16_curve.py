import sys, numpy as np, torch
sys.path.insert(0, ".")
from tests.helpers import load_golden
from tests.test_graph_cpu import make_config
from oracle import init as oinit, train as otrain
from phiseg_code_amd.phiseg import phiseg_model
g, cfg, var_order = load_golden("lidc_phiseg_bn")
cfg = dict(cfg, B=12)
params = otrain.make_params(var_order, cfg["weight_seed"], torch.float32, perturbed=False)
p0 = {k: v.detach().numpy() for k, v in params.items()}
batches = [oinit.synthetic_batch(cfg["B"], cfg["H"], cfg["nlabels"], 1000 + i) for i in range(8)]
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for dt, seed in (("f32", cfg["eps_seed"]), ("f32", cfg["eps_seed"] + 1), ("bf16", cfg["eps_seed"]), ("bf16", cfg["eps_seed"] + 1)):
    model = phiseg_model.phiseg(make_config(cfg, dt), rng_seed=seed)
    model.set_weights(p0)
    keys = sorted(model.loss_dict)
    rows = []
    for it in range(nsteps):
        x_np, s_np = batches[it % 8]
        out = model.sess.run([model.train_step] + [model.loss_dict[k] for k in keys], {model.x_inp: x_np, model.s_inp: s_np, model.training_pl: True, model.lr_pl: 1e-3})
        rows.append([float(v) for v in out[1:]])
    a = np.array(rows); i = keys.index("total_loss")
    print(dt, seed, "ELBO", [round(a[k:k + 50, i].mean(), 1) for k in range(0, nsteps, 50)])
    ce = [keys.index(k) for k in keys if k.startswith("residual")]
    print("   CE levels, last 50:", [round(a[-50:, c].mean(), 1) for c in ce], " KL sum last 50:", round(sum(a[-50:, keys.index(k)].mean() for k in keys if k.startswith("KL")), 1))
    del model
