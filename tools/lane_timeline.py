"""dev helper (GPU box): per-lane timeline of one replayed training step from in-graph wall-clock stamps (PHX_STAMPS=1).
rocprofv3's kernel trace serialises the lanes; these stamps (two 1-thread kernels per operator) do not.
usage: PHX_STAMPS=1 python tools/lane_timeline.py [bucket_us]"""
import os, sys
os.environ["PHX_STAMPS"] = "1"
sys.path.insert(0, ".")
import numpy as np, torch, bench
from phiseg_code_amd.data import synthetic
from phiseg_code_amd.phiseg import phiseg_model
cfg = bench.make_config(64, "bf16")
model = phiseg_model.phiseg(cfg)
sess = model.sess
plan = sess.plan_for([model.loss_tot], True, 64, True)
x, s = synthetic.make_batch(64, 128, cfg.nlabels, np.random.default_rng(0))
plan.set_input("x_input", x); plan.set_input("s_input", s)
sess.store.set_lr(1e-3)
for _ in range(6): plan.run()
plan.sync()
t = plan._stamp_buf.cpu().numpy().astype(np.int64)
rows = [(ph, name, lane, t[i] / 100.0, t[i + 1] / 100.0) for ph, name, lane, i in plan.stamps]      # us
t0 = min(r[3] for r in rows); t1 = max(r[4] for r in rows)
print("step span %.1f us over %d operators" % (t1 - t0, len(rows)))
nl = max(r[2] for r in rows) + 1
bucket = float(sys.argv[1]) if len(sys.argv) > 1 else 500.0
nb = int((t1 - t0) / bucket) + 1
busy = np.zeros((nl, nb))
for ph, name, lane, a, b in rows:
    a -= t0; b -= t0
    for k in range(int(a // bucket), int(b // bucket) + 1):
        lo, hi = max(a, k * bucket), min(b, (k + 1) * bucket)
        if hi > lo: busy[lane, k] += hi - lo
print("lane busy fraction per %.0f us bucket (operator begin..end, includes waiting inside an operator):" % bucket)
for ln in range(nl):
    print("lane %d  " % ln + " ".join("%3d" % int(100 * v / bucket) for v in busy[ln]))
# phase boundaries: first/last stamp of each (lane, phase, net)
def net(n): return n.split("/")[0]
agg = {}
for ph, name, lane, a, b in rows:
    k = (lane, ph, net(name))
    lo, hi, w = agg.get(k, (1e30, 0, 0.0))
    agg[k] = (min(lo, a - t0), max(hi, b - t0), w + (b - a))
for k in sorted(agg, key=lambda k: agg[k][0]):
    print("lane %d %s %-12s  %8.1f .. %8.1f us   busy %8.1f us" % (k[0], k[1], k[2], agg[k][0], agg[k][1], agg[k][2]))
if os.environ.get("PHX_STAMPS_DUMP"):
    with open(os.environ["PHX_STAMPS_DUMP"], "w") as f:
        for ph, name, lane, a, b in rows: f.write("%s\t%s\t%d\t%.2f\t%.2f\n" % (ph, name, lane, a - t0, b - t0))
