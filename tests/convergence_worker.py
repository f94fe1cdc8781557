"""Worker of the bf16-vs-fp32 convergence test (tests/test_model_gpu.py): trains phiseg_7_5 (n0 = 32, 128 x 128, batch norm, batch 12)
for N steps on ONE engine (f32 | bf16) with ONE Philox noise seed (the golden's seed + offset) and prints the first step's loss terms
and their means over the last TAIL steps as one JSON line; the test runs its four workers side by side (the fp32 engine's
deterministic filter gradients are single-block launches that leave the GPU idle: four processes overlap almost perfectly).  Run with PHX_DETERMINISTIC=1 the four trajectories are bit-reproducible (tests/test_deterministic_gpu.py),
so the comparison is a fixed number and not a draw from a chaotic system."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    nsteps, tail, dt, seed_off = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    torch.set_num_threads(min(8, torch.get_num_threads()))       # (four workers side by side; the host work is the weight initialiser)
    from oracle import init as oinit
    from oracle import train as otrain
    from phiseg_code_amd.phiseg import phiseg_model
    from tests.helpers import load_golden
    from tests.test_graph_cpu import make_config
    g, cfg, var_order = load_golden("lidc_phiseg_bn")
    cfg = dict(cfg, B=12)
    params = otrain.make_params(var_order, cfg["weight_seed"], torch.float32, perturbed=False)
    p0 = {k: v.detach().numpy() for k, v in params.items()}
    batches = [oinit.synthetic_batch(cfg["B"], cfg["H"], cfg["nlabels"], 1000 + i) for i in range(8)]
    model = phiseg_model.phiseg(make_config(cfg, dt), rng_seed=cfg["eps_seed"] + seed_off)
    model.set_weights(p0)
    keys = sorted(model.loss_dict)
    rows = []
    for it in range(nsteps):
        x_np, s_np = batches[it % len(batches)]
        out = model.sess.run([model.train_step] + [model.loss_dict[k] for k in keys],
                             {model.x_inp: x_np, model.s_inp: s_np, model.training_pl: True, model.lr_pl: 1e-3})
        rows.append([float(v) for v in out[1:]])
    a = np.array(rows)
    print("CONVERGENCE " + json.dumps(dict(keys=keys, dtype=dt, seed_offset=seed_off, finite=bool(np.isfinite(a).all()),
                                           first=a[0].tolist(), tail=a[-tail:].mean(axis=0).tolist())))


if __name__ == "__main__":
    main()
