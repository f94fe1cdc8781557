"""GPU box helper: the kernels of a rocprofv3 rocpd trace between the n-th dispatch of kernel A and the next dispatch of kernel B (by start
time, all streams): name, start and end relative to the window, duration, stream / queue id.
usage: python tools/trace_window.py <results.db> <A substring> <B substring> [n = -1: the last occurrence]"""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
A, Bn = sys.argv[2], sys.argv[3]
n = int(sys.argv[4]) if len(sys.argv) > 4 else -1
cols = [r[1] for r in db.execute("pragma table_info(rocpd_kernel_dispatch)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute("select s.kernel_name, d.start, d.end%s from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start"
                  % ((", d." + qcol) if qcol else "")).fetchall()
idx = [i for i, r in enumerate(rows) if A in r[0]]
i0 = idx[n]
i1 = next(i for i in range(i0 + 1, len(rows)) if Bn in rows[i][0])
t0 = rows[i0][1]
for r in rows[i0:i1 + 1]:
    print("%9.1f .. %9.1f  %7.1f us  q%-4s %s" % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3] if qcol else "", r[0][:110]))
