"""Instruction mix per kernel of the 3D detector from one rocprofv3 PMC pass (SQ_WAVES, SQ_INSTS_VALU, SQ_INSTS_SALU, SQ_INSTS_LDS, SQ_INSTS_VMEM_RD,
SQ_INSTS_VMEM_WR; --kernel-trace only): what a wave executes on average -- these kernels' waves share their SIMD with two or three others, so
the VALU count is time.  Usage: python scripts/pmc_det3d_insts.py counters.db"""
import sqlite3, statistics, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
by = {}
for k, c, v in rows:
    by.setdefault(k.replace("(anonymous namespace)::", "").split("(")[0], {}).setdefault(c, []).append(v)
cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"]
print(f"{'kernel':18s} {'calls':>5s} {'waves':>8s} " + " ".join(f"{c[9:]:>10s}" for c in cols[1:]) + "   (per wave: VALU SALU LDS VMEM_RD VMEM_WR)")
for k in sorted(by):
    if "rocclr" in k:
        continue
    m = {c: statistics.mean(by[k].get(c, [0])) for c in cols}
    w = max(m["SQ_WAVES"], 1.0)
    print(f"{k[:18]:18s} {len(by[k].get('SQ_WAVES', [])):5d} {m['SQ_WAVES']:8.0f} " + " ".join(f"{m[c]:10.0f}" for c in cols[1:]) +
          "   " + " ".join(f"{m[c] / w:7.1f}" for c in cols[1:]))
