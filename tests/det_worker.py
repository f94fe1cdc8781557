"""Worker of the deterministic-mode test: a few training steps, prints one digest of every loss term, gradient and parameter."""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    case, dtype, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # 0: the fixture's own batch
    from oracle import train as otrain
    from phiseg_code_amd.phiseg import phiseg_model
    from tests.helpers import golden_inputs, load_golden
    from tests.test_graph_cpu import make_config
    g, cfg, var_order = load_golden(case)
    if batch:
        cfg = dict(cfg, B=batch)
    model = phiseg_model.phiseg(make_config(cfg, dtype), rng_seed=cfg["eps_seed"])
    params, x, s = golden_inputs(cfg, var_order, dtype=torch.float64)
    model.set_weights({k: v.detach().numpy() for k, v in params.items()})
    h = hashlib.sha256()
    keys = sorted(model.loss_dict)
    for _ in range(steps):
        out = model.sess.run([model.train_step] + [model.loss_dict[k] for k in keys],
                             {model.x_inp: x, model.s_inp: s, model.training_pl: True, model.lr_pl: 1e-3})
        h.update(np.asarray(out[1:], dtype=np.float32).tobytes())
    st = model.sess.store
    for blob in (st.export(grads=True), st.export()):
        for k in sorted(blob):
            h.update(blob[k].tobytes())
    print("DIGEST", h.hexdigest(), float(out[-1]))


if __name__ == "__main__":
    main()
