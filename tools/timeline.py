"""dev helper: occupancy timeline of one training step from a rocprofv3 --kernel-trace CSV.
usage: python tools/timeline.py <kernel_trace.csv>
Splits the last step (between the last two k_step_increment launches) into intervals by the set of running kernels and reports
how much wall time has 0 / only narrow (< 256 work-groups) / at least one wide kernel in flight."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'k_step_increment' in r['Kernel_Name']]
a, b = marks[-2], marks[-1]
step = rows[a + 1:b + 1]
t0 = int(step[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in step)
ev = []
for r in step:
    wg = (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))) * max(1, int(r['Grid_Size_Y']) // max(1, int(r['Workgroup_Size_Y']))) * max(1, int(r['Grid_Size_Z']) // max(1, int(r['Workgroup_Size_Z'])))
    wide = wg >= 256
    ev.append((int(r['Start_Timestamp']), 1, wide, r['Kernel_Name']))
    ev.append((int(r['End_Timestamp']), -1, wide, r['Kernel_Name']))
ev.sort(key=lambda e: (e[0], e[1]))
nw = nn = 0; last = t0
acc = collections.Counter(); conc = collections.Counter()
for t, d, wide, _ in ev:
    dt = t - last
    if dt > 0:
        key = 'idle' if nw + nn == 0 else ('narrow-only' if nw == 0 else 'wide')
        acc[key] += dt; conc[min(nw + nn, 8)] += dt
    last = t
    if wide: nw += d
    else: nn += d
tot = t1 - t0
print(f"step wall {tot/1e6:.3f} ms, {len(step)} launches")
for k in ('idle', 'narrow-only', 'wide'): print(f"  {k:12s} {acc[k]/1e6:7.3f} ms  {100*acc[k]/tot:5.1f}%")
print("  kernels in flight -> ms:", {k: round(v / 1e6, 2) for k, v in sorted(conc.items())})
dur = collections.Counter(); cnt = collections.Counter()
for r in step:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    b_ = '<8us' if d < 8000 else '<16us' if d < 16000 else '<32us' if d < 32000 else '<64us' if d < 64000 else '>=64us'
    dur[b_] += d; cnt[b_] += 1
for k in ('<8us', '<16us', '<32us', '<64us', '>=64us'): print(f"  kernels {k:6s}: {cnt[k]:5d} launches, {dur[k]/1e6:6.2f} ms summed")
