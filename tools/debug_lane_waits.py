"""dev helper (GPU box): list the event records / waits of the captured launch list around the start of each lane."""
import os, sys
sys.path.insert(0, ".")
import numpy as np, torch, bench
from phiseg_code_amd.phiseg import phiseg_model
cfg = bench.make_config(64, "bf16")
model = phiseg_model.phiseg(cfg)
plan = model.sess.plan_for([model.loss_tot], True, 64, True)
L = plan.L
lanes = {int(s.value): i for i, s in enumerate(plan._lanes)}
evname = {}
def lane_of_args(args):
    for a in args[::-1]:
        v = getattr(a, "value", a)
        if isinstance(v, int) and v in lanes: return lanes[v]
    return -1
first_kernel = {}
for i, (f, a) in enumerate(plan.launches):
    nm = getattr(f, "__name__", str(f))
    if f is L.event_record:
        evname[a[0].value] = (i, lanes[int(a[1].value)])
    ln = lane_of_args(a)
    if f is L.stream_wait_event:
        src = evname.get(a[1].value, (None, None))
        if ln in (1, 2) and len([1 for k in first_kernel if k == ln]) == 0 or (ln == 1 and i < first_kernel.get(1, 0) + 400):
            print("%5d  lane %d WAITS for event recorded at %s on lane %s" % (i, ln, src[0], src[1]))
    elif f is not L.event_record and nm != "_noop":
        if ln not in first_kernel:
            first_kernel[ln] = i
            print("%5d  lane %d first launch %s" % (i, ln, nm))
print("n launches", len(plan.launches), "fwd", plan.n_launch_fwd)
names = [(op.name, plan.op_lane[op]) for op in plan.ops]
pri = [k for k, (n, l) in enumerate(names) if n.startswith("prior/")]
print("prior ops at graph positions %d..%d of %d" % (pri[0], pri[-1], len(names)))
print([n for n, l in names[pri[0]:pri[0] + 4]])
op0 = plan.ops[pri[0]]
print("first prior op inputs:", [(t.op.name, plan.op_lane.get(t.op)) for t in op0.inputs], "real producers:", [getattr(plan._real_producer(t), "name", None) for t in op0.inputs])
