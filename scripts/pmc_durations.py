"""Kernel durations recorded in a rocprofv3 counter-collection database (how much the counter pass itself stretches a
dispatch).  Usage (GPU box): python scripts/pmc_durations.py <results.db> [last_n]"""
import json, sqlite3, statistics, sys
db = sqlite3.connect(sys.argv[1])
last = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rows = db.execute("select name, start, end from kernels order by start").fetchall()
by = {}
for n, s, e in rows:
    by.setdefault(n.split("(")[0], []).append((e - s) / 1e3)
for k, v in sorted(by.items()):
    v = v[-last:]
    print(json.dumps({"kernel": k, "dispatches": len(v), "median_us": round(statistics.median(v), 2)}))
