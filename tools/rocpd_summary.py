"""Summarise a rocprofv3 rocpd database (the default output format of ROCm 7.2's rocprofv3) as CSV:
   python tools/rocpd_summary.py <results.db> [--pmc]  -> kernel, calls, total_us, avg_us, min_us, max_us [, counter sums / call]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    names = dict(cur.execute("select id, kernel_name from rocpd_info_kernel_symbol").fetchall())
    rows = cur.execute("select id, kernel_id, start, end from rocpd_kernel_dispatch").fetchall()
    pmc = {}
    if "--pmc" in sys.argv:
        cols = [r[1] for r in cur.execute("pragma table_info(rocpd_pmc_event)")]
        info = dict(cur.execute("select id, name from rocpd_info_pmc").fetchall())
        ev2disp = dict(cur.execute("select id, dispatch_id from rocpd_kernel_dispatch").fetchall())
        key = "event_id" if "event_id" in cols else "id"
        # pmc events reference the dispatch through the event id shared with rocpd_kernel_dispatch.event_id when present
        dcols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
        if "event_id" in dcols:
            ev2k = dict(cur.execute("select event_id, kernel_id from rocpd_kernel_dispatch").fetchall())
        else:
            ev2k = {}
        for eid, pid, val in cur.execute("select %s, pmc_id, value from rocpd_pmc_event" % key):
            k = ev2k.get(eid)
            if k is None:
                continue
            d = pmc.setdefault(names.get(k, str(k)), {})
            d[info[pid]] = d.get(info[pid], 0.0) + float(val)
    agg = {}
    for _id, kid, s, e in rows:
        a = agg.setdefault(names.get(kid, str(kid)), [0, 0.0, 1e30, 0.0])
        us = (e - s) / 1000.0
        a[0] += 1; a[1] += us; a[2] = min(a[2], us); a[3] = max(a[3], us)
    counters = sorted({c for d in pmc.values() for c in d})
    print(",".join(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us"] + [c + "_per_call" for c in counters]))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        extra = ["%.1f" % (pmc.get(k, {}).get(c, 0.0) / a[0]) for c in counters]
        print(",".join(['"%s"' % k, str(a[0]), "%.1f" % a[1], "%.2f" % (a[1] / a[0]), "%.2f" % a[2], "%.2f" % a[3]] + extra))


if __name__ == "__main__":
    main()
