"""Lane timeline of one training step from a rocprofv3 rocpd database: per-stream busy time, the time both lanes run, idle gaps.
usage: prof_timeline.py results.db [nsteps_back]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)").fetchall()]
key = "stream_id" if "stream_id" in cols else "queue_id"
rows = cur.execute("select d.start, d.end, d.%s, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                   "on d.kernel_id=s.id order by d.start" % key).fetchall()
adam = [i for i, r in enumerate(rows) if "k_adam" in r[3]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lo, hi = adam[-back - 1] + 1, adam[-back] + 1            # one whole step: after an optimizer launch up to and including the next
step = rows[lo:hi]
t0, t1 = step[0][0], max(r[1] for r in step)
print("step: %d launches, %.3f ms wall (first start -> last end)" % (len(step), (t1 - t0) / 1e6))
lanes = defaultdict(list)
for r in step:
    lanes[r[2]].append(r)

def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0)

for k, v in sorted(lanes.items(), key=lambda kv: -len(kv[1])):
    print("lane %-6s launches %4d  busy %.3f ms  kernel-time sum %.3f ms" % (k, len(v), union([(r[0], r[1]) for r in v]) / 1e6,
                                                                          sum(r[1] - r[0] for r in v) / 1e6))
allu = union([(r[0], r[1]) for r in step])
print("any lane busy %.3f ms; nothing running %.3f ms" % (allu / 1e6, (t1 - t0 - allu) / 1e6))
# time with exactly one / two or more kernels running, and what runs alone
ev = []
for r in step:
    ev.append((r[0], 1, r[3])); ev.append((r[1], -1, r[3]))
ev.sort()
run, last, alone, multi = {}, t0, defaultdict(float), 0.0
for t, d, name in ev:
    n = sum(run.values())
    if n == 1:
        alone[[k for k, c in run.items() if c][0]] += t - last
    elif n >= 2:
        multi += t - last
    run[name] = run.get(name, 0) + d
    last = t
print("two or more kernels running %.3f ms; exactly one %.3f ms" % (multi / 1e6, sum(alone.values()) / 1e6))
print("kernels running ALONE (ms/step):")
for name, t in sorted(alone.items(), key=lambda kv: -kv[1])[:25]:
    print("  %8.3f  %s" % (t / 1e6, name[:110]))
# gaps > 3 us with nothing running: which kernel follows
gaps = []
iv = sorted((r[0], r[1], r[3]) for r in step)
ce = iv[0][1]
for s, e, name in iv[1:]:
    if s > ce + 3000:
        gaps.append((s - ce, name))
    ce = max(ce, e)
print("idle gaps > 3 us: %d, total %.3f ms" % (len(gaps), sum(g[0] for g in gaps) / 1e6))
for g, name in sorted(gaps, reverse=True)[:10]:
    print("  %7.1f us before %s" % (g / 1e3, name[:100]))
