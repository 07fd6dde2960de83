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

if "--json" in sys.argv:
    path = sys.argv[sys.argv.index("--json") + 1]
    commit = sys.argv[sys.argv.index("--commit") + 1] if "--commit" in sys.argv else None
    tag = sys.argv[sys.argv.index("--tag") + 1] if "--tag" in sys.argv else "x"
    vs = by.get(("void k_mid<4, 0>", "SQ_VALU_MFMA_BUSY_CYCLES"), [])[-last:]
    if vs:
        json.dump({"kernel": "void k_mid<4, 0>", "sq_valu_mfma_busy_cycles_median": statistics.median(vs), "simds": 1024, "clock_ghz": 2.4,
                   "_commit": commit,
                   "note": f"rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES over bench.py --timed-only, last {last} dispatches (profiles/{tag}_pmc_mfma.txt); "
                           "utilisation = busy / (simds x launch cycles)"}, open(path, "w"), indent=1)
