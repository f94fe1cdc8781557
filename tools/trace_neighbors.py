"""GPU box helper: which kernels surround a given kernel in a rocprofv3 rocpd trace (by start time)?"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); pat = sys.argv[2]
rows = db.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start").fetchall()
prev, nxt = collections.Counter(), collections.Counter()
for i, r in enumerate(rows):
    if pat in r[0]:
        if i > 0: prev[rows[i - 1][0][:60]] += 1
        if i + 1 < len(rows): nxt[rows[i + 1][0][:60]] += 1
print("before:", prev.most_common(6)); print("after:", nxt.most_common(6))
