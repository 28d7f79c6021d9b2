"""HBM traffic of the build kernels from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md,
'rocprofv3 PMC slots'):  python tools/build_traffic.py <fetch.db> <write.db> <config> <out.json>
Sums, per linearisation, the build kernels (k_build*, k_long_reduce, k_schur_*) and applies the guide's gfx950 correction: FETCH_SIZE
tallies a 128-B request of a wide coalesced stream as 64 B, so the read side is doubled (an upper estimate for narrow streams)."""
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_source_sha():
    """what the build kernels' traffic depends on: bench.py quotes a traffic file only while this is unchanged"""
    h = hashlib.sha256()
    for f in ("large-scale-sfm_amd/csrc/kernels.hip", "large-scale-sfm_amd/csrc/plan.cpp"):
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    names = dict(cur.execute("select id, kernel_name from rocpd_info_kernel_symbol").fetchall())
    info = dict(cur.execute("select id, name from rocpd_info_pmc").fetchall())
    ev2k = dict(cur.execute("select event_id, kernel_id from rocpd_kernel_dispatch").fetchall())
    calls, tot = {}, {}
    for _id, kid in cur.execute("select id, kernel_id from rocpd_kernel_dispatch"):
        calls[names[kid]] = calls.get(names[kid], 0) + 1
    for eid, pid, val in cur.execute("select event_id, pmc_id, value from rocpd_pmc_event"):
        if info[pid] != counter or eid not in ev2k:
            continue
        k = names[ev2k[eid]]
        tot[k] = tot.get(k, 0.0) + float(val)
    return {k: (tot.get(k, 0.0) / calls[k], calls[k]) for k in calls}


def main():
    fetch, write, config, out = sys.argv[1:5]
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    is_build = lambda k: any(t in k for t in ("k_build", "k_long_reduce", "k_schur_"))
    rows = {}
    lin = max(c for k, (_, c) in f.items() if is_build(k))              # linearisations in the run = launches of the busiest build kernel
    fk = wk = 0.0
    for k in sorted(f):
        if not is_build(k):
            continue
        fkb, cf = f[k]; wkb, cw = w.get(k, (0.0, 0))
        rows[k] = {"launches": cf, "FETCH_SIZE_KB_per_launch": fkb, "WRITE_SIZE_KB_per_launch": wkb}
        fk += fkb * cf / lin; wk += wkb * cw / lin
    raw = (fk + wk) * 1024.0
    json.dump({"workload": config + " 1 GPU", "kernels": rows, "linearisations": lin, "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk,
               "hbm_bytes_raw": raw, "hbm_bytes_gfx950_corrected": (2.0 * fk + wk) * 1024.0,
               "build_source_sha": build_source_sha(),
               "build_source_files": "sha256 over csrc/kernels.hip + csrc/plan.cpp, first 16 hex digits (bench.py quotes this file only while they are unchanged)",
               "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 "
                       "--config %s`; per linearisation = summed over the build kernels; MI355X_MICROARCH.md: FETCH_SIZE on gfx950 counts 64 B per 128-B "
                       "request of a wide coalesced stream, hence the doubled read side; narrow (4-8 B/lane) streams are uncalibrated, so the corrected "
                       "figure is an upper estimate" % config}, open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
