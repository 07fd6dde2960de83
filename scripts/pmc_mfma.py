"""Median of a rocprofv3 PMC counter per kernel over the last N dispatches of a counter-collection database.
Usage (on the GPU box, the databases are too big to travel): python scripts/pmc_mfma.py <results.db> [last_n]"""
import json, sqlite3, statistics, sys
db = sqlite3.connect(sys.argv[1])
last = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rows = db.execute("select kernel_name, counter_name, value from counters_collection order by start").fetchall()
by = {}
for k, c, v in rows:
    by.setdefault((k.split("(")[0], c), []).append(v)
for (k, c), vs in sorted(by.items()):
    vs = vs[-last:]
    print(json.dumps({"kernel": k, "counter": c, "dispatches": len(vs), "median": statistics.median(vs), "mean": sum(vs) / len(vs)}))
