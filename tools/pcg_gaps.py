"""Durations of the PCG kernels and the idle gaps between consecutive dispatches, from a rocprofv3 rocpd database:
   python tools/pcg_gaps.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
names = dict(cur.execute("select id, kernel_name from rocpd_info_kernel_symbol").fetchall())
rows = sorted(cur.execute("select start, end, kernel_id from rocpd_kernel_dispatch").fetchall())
stats = {}
prev = None
for s, e, k in rows:
    n = names.get(k, str(k))
    short = "spmv" if "k_pcg_spmv" in n else "update" if "k_pcg_update" in n else None
    if short and prev and prev[2] in ("spmv", "update") and (e - s) > 2000:
        st = stats.setdefault(prev[2] + "->" + short, [0, 0.0, 0.0])
        st[0] += 1; st[1] += (e - s) / 1000.0; st[2] += (s - prev[1]) / 1000.0
    prev = (s, e, short)
for k, (n, dur, gap) in stats.items():
    print("%-16s n=%5d  kernel %.2f us  gap before %.2f us" % (k, n, dur / n, gap / n))
