"""Debug (GPU box): multi-lane hipGraph capture on the tiny model."""
import sys, faulthandler
faulthandler.enable()
sys.path.insert(0, ".")
import numpy as np
from tests.test_model_gpu import build
from phiseg_code_amd import engine

n_lanes = int(sys.argv[1])
cut = int(sys.argv[2]) if len(sys.argv) > 2 else -1
g, cfg, var_order, model, params, x_np, s_np = build("tiny_phiseg_bn")
store = model.sess._ensure_store()
if cut == 0:      # forward only
    plan = engine.Plan(store, list(model.s_out_list) + [model.loss_tot], loss=None, batch=cfg["B"], training=True, compute_dtype="f32",
                       rng_seed=42, n_lanes=n_lanes)
else:
    plan = engine.Plan(store, [model.loss_tot], loss=model.loss_tot, batch=cfg["B"], training=True, compute_dtype="f32",
                       optimize=True, rng_seed=42, n_lanes=n_lanes)
plan.set_input("x_input", x_np); plan.set_input("s_input", s_np)
print("launches", len(plan.launches), "events", len(plan._events), flush=True)
recorded, bad = {}, 0
lane_of_stream = {st.value: i for i, st in enumerate(plan._lanes)}
last_rec_on_lane = {}
for idx, (fn, args) in enumerate(plan.launches):
    nm = getattr(fn, "__name__", str(fn))
    if nm == "phx_event_record":
        recorded[args[0].value] = (idx, lane_of_stream[args[1].value])
    elif nm == "phx_stream_wait_event":
        if args[1].value not in recorded:
            bad += 1
            print("WAIT BEFORE RECORD at", idx, "lane", lane_of_stream[args[0].value], flush=True)
print("event order check: bad =", bad, flush=True)
plan.run_eager(); plan.sync(); print("eager ok", float(plan.fetch(model.loss_tot)), flush=True)
plan._warm = True
plan.run(); plan.sync(); print("capture ok", float(plan.fetch(model.loss_tot)), flush=True)
plan.run(); plan.sync(); print("replay ok", float(plan.fetch(model.loss_tot)), flush=True)
