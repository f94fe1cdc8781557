"""GPU box: which variables' gradients differ between two identical runs (PHX_DETERMINISTIC as set in the environment)?"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from phiseg_code_amd.phiseg import phiseg_model
from tests.helpers import golden_inputs, load_golden
from tests.test_graph_cpu import make_config
case, dtype = sys.argv[1], sys.argv[2]
g, cfg, var_order = load_golden(case)
def run():
    model = phiseg_model.phiseg(make_config(cfg, dtype), rng_seed=cfg["eps_seed"])
    params, x, s = golden_inputs(cfg, var_order, dtype=torch.float64)
    model.set_weights({k: v.detach().numpy() for k, v in params.items()})
    keys = sorted(model.loss_dict)
    out = model.sess.run([model.train_step] + [model.loss_dict[k] for k in keys], {model.x_inp: x, model.s_inp: s, model.training_pl: True, model.lr_pl: 0.0})
    return dict(zip(keys, [float(v) for v in out[1:]])), model.sess.store.export(grads=True)
la, ga = run(); lb, gb = run()
print({k: (la[k], lb[k]) for k in la if la[k] != lb[k]})
for k in ga:
    if not np.array_equal(ga[k], gb[k]):
        print("DIFF", k, np.abs(ga[k] - gb[k]).max(), np.abs(ga[k]).max())
