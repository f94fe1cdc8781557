"""Debug (GPU box): capture only a prefix of the multi-lane launch list (plus a join) to bisect the EndCapture crash."""
import sys, ctypes, faulthandler
faulthandler.enable()
sys.path.insert(0, ".")
from tests.test_model_gpu import build
from phiseg_code_amd import engine
cut = int(sys.argv[1])
g, cfg, var_order, model, params, x_np, s_np = build("tiny_phiseg_bn")
store = model.sess._ensure_store()
plan = engine.Plan(store, [model.loss_tot], loss=model.loss_tot, batch=cfg["B"], training=True, compute_dtype="f32",
                   optimize=False, rng_seed=42, n_lanes=3)
plan.set_input("x_input", x_np); plan.set_input("s_input", s_np)
L = plan.L
n = len(plan.launches)
if cut < 0:
    print("launches", n, "fwd", plan.n_launch_fwd, flush=True); sys.exit(0)
plan.run_eager(); plan.sync()
names = [getattr(fn, "__name__", "?") for fn, a in plan.launches]
lane_of = {st.value: i for i, st in enumerate(plan._lanes)}
def desc(i):
    fn, a = plan.launches[i]
    nm = names[i]
    if nm == "phx_event_record": return "rec(ev%x)@L%d" % (a[0].value & 0xffff, lane_of[a[1].value])
    if nm == "phx_stream_wait_event": return "L%d.wait(ev%x)" % (lane_of[a[0].value], a[1].value & 0xffff)
    st = a[-1]
    return "%s@L%d" % (nm[4:], lane_of.get(getattr(st, "value", st), -1))
if len(sys.argv) > 2:
    for i in range(int(sys.argv[2]), cut): print(i, desc(i), flush=True)
    # where was the event of the last wait recorded, and what did its lane / the waiting lane do around it?
    widx = max(i for i in range(cut) if names[i] == "phx_stream_wait_event")
    ev = plan.launches[widx][1][1].value
    ridx = [i for i in range(cut) if names[i] == "phx_event_record" and plan.launches[i][1][0].value == ev]
    print("last wait at", widx, desc(widx), "recorded at", ridx, [desc(i) for i in ridx], flush=True)
    for i in range(max(0, ridx[0] - 6), ridx[0] + 3): print("  R", i, desc(i), flush=True)
    wl = lane_of[plan.launches[widx][1][0].value]
    prev = [i for i in range(widx) if desc(i).endswith("@L%d" % wl) or desc(i).startswith("L%d." % wl)][-8:]
    for i in prev: print("  W-lane history", i, desc(i), flush=True)
print("cut", cut, "last op", names[cut - 1], flush=True)
L.graph_begin_capture(plan._lanes[0])
for fn, args in plan.launches[:cut]:
    fn(*args)
for ln in range(1, 3):
    ev = ctypes.c_void_p(); L.event_create(ctypes.byref(ev))
    L.event_record(ev, plan._lanes[ln]); L.stream_wait_event(plan._lanes[0], ev)
ge = ctypes.c_void_p()
L.graph_end_capture(plan._lanes[0], ctypes.byref(ge))
print("cut", cut, "OK", flush=True)
