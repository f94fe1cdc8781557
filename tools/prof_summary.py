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
