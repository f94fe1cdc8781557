"""dev helper: per-(kernel, grid) duration of the last step in two rocprofv3 kernel traces (A/B of two builds).
usage: python tools/compare_traces.py a.csv b.csv [name-filter]"""
import csv, sys, collections
def load(f):
    rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if 'k_step_increment' in r['Kernel_Name']]
    d = collections.Counter(); c = collections.Counter()
    for s in range(2, len(marks)):                      # skip the first (eager / capture) steps
        for r in rows[marks[s - 1] + 1:marks[s] + 1]:
            k = (r['Kernel_Name'].split('(')[0][:60], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Z']))
            d[k] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); c[k] += 1
    n = max(1, len(marks) - 2)
    return {k: (v / n / 1e3, c[k] / n) for k, v in d.items()}
a, b = load(sys.argv[1]), load(sys.argv[2]); flt = sys.argv[3] if len(sys.argv) > 3 else ''
ta = tb = 0
for k in sorted(a, key=lambda k: -a[k][0]):
    if flt not in k[0] or k not in b: continue
    ta += a[k][0]; tb += b[k][0]
    if a[k][0] > 40: print(f"{k[0]:62s} wg={k[1]:6d} z={k[2]} n={a[k][1]:5.1f}  {a[k][0]:8.1f} -> {b[k][0]:8.1f} us  {100*(b[k][0]/a[k][0]-1):+5.1f}%")
print(f"total {ta/1e3:.3f} -> {tb/1e3:.3f} ms")
