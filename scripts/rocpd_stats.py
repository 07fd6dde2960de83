"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table
(name, calls, total/avg/min/max us).  Optionally restrict to the last N dispatches of each
kernel (steady state).  Usage: python scripts/rocpd_stats.py results.db [--last N] > profiles/x.txt"""
import sqlite3, sys, statistics
db = sqlite3.connect(sys.argv[1])
last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else None
rows = db.execute("select name, start, end, grid_x, grid_y, workgroup_x, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size from kernels order by start").fetchall()
by = {}
for r in rows:
    by.setdefault(r[0], []).append(r)
print(f"{'kernel':60s} {'calls':>7s} {'total_us':>12s} {'avg_us':>9s} {'med_us':>9s} {'min_us':>9s} {'max_us':>9s}  grid wg vgpr agpr sgpr lds scratch")
tot = 0
out = []
for name, rs in by.items():
    if last: rs = rs[-last:]
    d = [(r[2] - r[1]) / 1e3 for r in rs]
    out.append((sum(d), name, len(d), sum(d) / len(d), statistics.median(d), min(d), max(d), rs[-1]))
    tot += sum(d)
for t, name, c, a, med, mn, mx, r in sorted(out, reverse=True):
    print(f"{name[:60]:60s} {c:7d} {t:12.1f} {a:9.2f} {med:9.2f} {mn:9.2f} {mx:9.2f}  {r[3]}x{r[4]} {r[5]} {r[6]} {r[7]} {r[8]} {r[9]} {r[10]}")
print(f"total kernel time {tot:.1f} us over {len(rows)} dispatches" + (f" (last {last} per kernel)" if last else ""))

if "--json" in sys.argv:
    import json
    path = sys.argv[sys.argv.index("--json") + 1]
    commit = sys.argv[sys.argv.index("--commit") + 1] if "--commit" in sys.argv else None
    js = {name.split("(")[0]: round(a, 3) for t, name, c, a, med, mn, mx, r in out}
    js["_commit"] = commit                     # the tree these averages were measured on (tests/test_profiles_gpu.py checks it against the built library)
    js["_last"] = last
    json.dump(js, open(path, "w"), indent=1)
