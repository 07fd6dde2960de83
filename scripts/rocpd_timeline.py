"""Kernel timeline excerpt from a rocprofv3 rocpd database: start / end (us, relative) of consecutive dispatches.
Usage: python scripts/rocpd_timeline.py <results.db> [first_index] [count]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
count = int(sys.argv[3]) if len(sys.argv) > 3 else 24
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
qcol = "queue_id" if "queue_id" in cols else None
rows = db.execute(f"select s.kernel_name, d.start, d.end{', d.' + qcol if qcol else ''} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start limit {count} offset {first}").fetchall()
t0 = rows[0][1]
prev_end = t0
for r in rows:
    name = r[0].split("(")[0].replace("(anonymous namespace)::", "")[:28]
    print(f"{name:28s} q={r[3] if qcol else '-':>3} start {(r[1]-t0)/1e3:9.2f}  end {(r[2]-t0)/1e3:9.2f}  dur {(r[2]-r[1])/1e3:7.2f}  gap_from_prev_end {(r[1]-prev_end)/1e3:7.2f}")
    prev_end = max(prev_end, r[2])
