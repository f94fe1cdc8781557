"""Where the reference's own loop (sess.run with a loss fetch per step) spends its host time: set_input / run + sync / fetch."""
import time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from phiseg_code_amd.phiseg import phiseg_model
from phiseg_code_amd.data import synthetic
cfg = bench.make_config(64, "bf16")
model = phiseg_model.phiseg(cfg)
x, s = synthetic.philox_batch(64, 128, cfg.nlabels, seed=4321)
fd = {model.x_inp: x, model.s_inp: s, model.training_pl: True, model.lr_pl: 1e-3}
for _ in range(4):
    model.sess.run([model.train_step, model.loss_tot], fd)
plan = list(model.sess.plans.values())[0]
print("x", x.dtype, x.shape, "s", s.dtype, s.shape)
N = 30
def T(f):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N): f()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / N
print("set_input x   %.3f ms" % T(lambda: plan.set_input("x_input", x)))
print("set_input s   %.3f ms" % T(lambda: plan.set_input("s_input", s)))
print("run + sync    %.3f ms" % T(lambda: plan.run(sync=True)))
print("fetch loss    %.3f ms" % T(lambda: plan.fetch(model.loss_tot)))
print("sess.run      %.3f ms" % T(lambda: model.sess.run([model.train_step, model.loss_tot], fd)))
xp = torch.empty(x.shape, dtype=torch.float32).pin_memory()
def pinned():
    xp.numpy()[...] = x
    plan.L.memcpy_h2d(plan.feeds["x_input"].ptr, xp.data_ptr(), x.nbytes, plan.stream)
    plan.L.stream_sync(plan.stream)
print("pinned path x %.3f ms" % T(pinned))
def pinned_nocopy():
    plan.L.memcpy_h2d(plan.feeds["x_input"].ptr, xp.data_ptr(), x.nbytes, plan.stream)
    plan.L.stream_sync(plan.stream)
print("pinned dma x  %.3f ms" % T(pinned_nocopy))
