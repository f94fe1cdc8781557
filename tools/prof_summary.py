"""Summarise a rocprofv3 rocpd sqlite database: per-kernel count / total / average duration."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cur = db.cursor()
rows = cur.execute("select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3 "
                   "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id "
                   "group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("total kernel time %.2f ms over %g steps = %.2f ms/step, %d launches/step" % (tot, steps, tot / steps, sum(r[1] for r in rows) / steps))
print("%-100s %9s %11s %7s %10s" % ("kernel", "calls/st", "ms/step", "%", "avg us"))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print("%-100s %9.1f %11.3f %6.1f%% %10.1f" % (r[0][:100], r[1] / steps, r[2] / steps, 100 * r[2] / tot, r[3]))

# forward + data-gradient convolution launches in situ (the family bench.py's roofline times one launch at a time): for roofline.frac_in_situ
import json
fam = [r for r in rows if any(t in r[0] for t in ("k_conv3x3_pp", "k_conv3x3_c32", "k_conv3x3_mfma"))]
info = {"conv_fwd_dgrad_ms_per_step": sum(r[2] for r in fam) / steps, "conv_fwd_dgrad_launches_per_step": sum(r[1] for r in fam) / steps,
        "kernel_ms_per_step": tot / steps, "launches_per_step": sum(r[1] for r in rows) / steps, "steps": steps,
        "source": "rocprofv3 --kernel-trace --stats -- python bench.py --steps 8 --warmup 3 (tools/collect_profiles.sh)"}
print("conv fwd + dgrad in situ: %.3f ms/step over %.0f launches" % (info["conv_fwd_dgrad_ms_per_step"], info["conv_fwd_dgrad_launches_per_step"]))
if len(sys.argv) > 4:
    json.dump(info, open(sys.argv[4], "w"), indent=1)
