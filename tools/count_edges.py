"""Dev (GPU box): launches / event records / cross-lane waits of the bench plan, per lane."""
import sys, types, importlib, collections
sys.path.insert(0, ".")
import numpy as np
from bench import make_config
from phiseg_code_amd.phiseg import phiseg_model
cfg = make_config(64, "bf16")
model = phiseg_model.phiseg(cfg)
plan = model.sess.plan_for([model.loss_tot], True, 64, True)
L = plan.L
names = collections.Counter()
lane_of = {id(s): i for i, s in enumerate(plan._lanes)}
per_lane = collections.Counter()
for fn, args in plan.launches + plan.opt_launches:
    n = getattr(fn, "__name__", str(fn))
    names[n] += 1
waits = [a for f, a in plan.launches if getattr(f, "__name__", "") == "phx_stream_wait_event"]
recs = [a for f, a in plan.launches if getattr(f, "__name__", "") == "phx_event_record"]
print("launch entries", len(plan.launches), "+ opt", len(plan.opt_launches), "| event records", len(recs), "| waits", len(waits))
print(names.most_common(12))
