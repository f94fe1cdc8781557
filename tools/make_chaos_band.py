"""tests/golden/lidc_chaos_band.npz: the oracle's own run-to-run drift at n0 = 32, 128 x 128, batch 2 -- the relative difference of
the 8-step loss trajectory when the input is perturbed by 1e-6 (oracle/train.py, fp32 torch-CPU; TF1 Adam moves every weight by
~lr * sign-like(m / sqrt(v)), so two exact implementations drift apart).  tests/test_model_gpu.py::lidc_trajectory uses it as the
tolerance band of the HIP trajectory instead of recomputing the second oracle trajectory (95 s) in every test session.
Oracle only -- nothing of /root/reference is read."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle import init as oinit          # noqa: E402
from oracle import train as otrain        # noqa: E402
from tests.helpers import load_golden     # noqa: E402

NSTEP, LR = 8, 2e-5
g, cfg, var_order = load_golden("lidc_phiseg_bn")
x_np, s_np = oinit.synthetic_batch(cfg["B"], cfg["H"], cfg["nlabels"], cfg["data_seed"])
out = []
for scale in (np.float32(1.0), np.float32(1.000001)):
    params = otrain.make_params(var_order, cfg["weight_seed"], torch.float32, perturbed=True)
    out.append([l["total_loss"] for l in otrain.train_steps(params, [(x_np * scale, s_np)], cfg, cfg["eps_seed"], lr=LR, n_steps=NSTEP,
                                                            dtype=torch.float32)])
ref, ref2 = np.array(out[0]), np.array(out[1])
chaos = np.abs(ref2 - ref) / np.abs(ref)
np.savez(os.path.join(ROOT, "tests", "golden", "lidc_chaos_band.npz"), chaos=chaos, ref=ref, nstep=NSTEP, lr=LR)
print("chaos band:", chaos)
