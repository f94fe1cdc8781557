"""Debug (GPU box): reproducibility of the ELBO across repeated runs / different fetch sets at the bench configuration."""
import importlib, types, sys
import numpy as np
sys.path.insert(0, ".")
from oracle import init as oinit
from phiseg_code_amd.phiseg import phiseg_model
base = importlib.import_module("phiseg_code_amd.phiseg.experiments.phiseg_7_5")
cfg = types.SimpleNamespace(**{k: getattr(base, k) for k in dir(base) if not k.startswith("_")})
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg.batch_size, cfg.compute_dtype = B, (sys.argv[2] if len(sys.argv) > 2 else "bf16")
x, s = oinit.synthetic_batch(B, 128, cfg.nlabels, 77)
model = phiseg_model.phiseg(cfg, rng_seed=5)
keys = sorted(model.loss_dict)
fd = {model.x_inp: x, model.s_inp: s, model.training_pl: True, model.lr_pl: 1e-3}
for i in range(3):
    print("A", i, float(model.sess.run(model.loss_tot, fd)))
for i in range(3):
    tot, terms = model.sess.run([model.loss_tot, [model.loss_dict[k] for k in keys]], fd)
    print("B", i, float(tot), [round(float(v), 1) for v in terms][:6])
for i in range(2):
    print("A", i, float(model.sess.run(model.loss_tot, fd)))
