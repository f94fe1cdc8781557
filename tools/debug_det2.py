"""GPU box: first op (in graph order) whose output differs between two identical forward evaluations."""
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
    ops = [op for op in model.graph.ops if op.type in ("conv_unit", "avgpool", "bilinear_up", "concat", "global_avgpool", "tile_pixels", "add")
           and (op.name.startswith("posterior") or op.name.startswith("likelihood") or op.name.startswith("prior"))]
    # only the training instances (first occurrence of each name prefix)
    ts = [op.outputs[0] for op in ops]
    need = model.loss_tot
    vals = model.sess.run(ts[:120] + [need], {model.x_inp: x, model.s_inp: s, model.training_pl: True})
    return [op.name for op in ops[:120]], vals
na, va = run(); nb, vb = run()
n = 0
for name, a, b in zip(na, va, vb):
    if not np.array_equal(a, b):
        print("DIFF %-50s shape %s max|d| %.3e of %.3e" % (name, a.shape, np.abs(a - b).max(), np.abs(a).max()))
        n += 1
        if n > 8: break
print("loss", va[-1], vb[-1])
