"""One damping trial as the GPU saw it, from a rocprofv3 rocpd database: every dispatch between two k_build launches with its
start offset, duration and the idle gap before it:   python tools/trial_timeline.py <results.db> [trial index, default 6] [marker kernel, default k_build;
k_zero for scenes whose build is several launches]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
names = dict(cur.execute("select id, kernel_name from rocpd_info_kernel_symbol").fetchall())
rows = sorted(cur.execute("select start, end, kernel_id from rocpd_kernel_dispatch").fetchall())
short = lambda n: n.split("(")[0].replace("miba::", "")[:60]
marker = sys.argv[3] if len(sys.argv) > 3 else "k_build"
starts = [i for i, (s, e, k) in enumerate(rows) if marker in names.get(k, "")]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 6
a, b = starts[which], starts[which + 1]
t0 = rows[a][0]
prev_end = t0
busy = 0.0
agg = {}
for s, e, k in rows[a:b]:
    n = short(names.get(k, str(k)))
    if "pcg_iter" in n or "rocclr" in n or True:
        ag = agg.setdefault(n, [0, 0.0, 0.0])
        ag[0] += 1; ag[1] += (e - s) / 1000.0; ag[2] += max(0, s - prev_end) / 1000.0
    busy += (e - s) / 1000.0
    prev_end = e
print("trial %d: %.1f us wall, %.1f us busy" % (which, (rows[b][0] - t0) / 1000.0, busy))
for n, (c, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-62s x%3d  %8.1f us  idle before %7.1f us" % (n, c, d, g))
