"""Per-kernel HBM traffic of the detectors from the two rocprofv3 PMC passes of scripts/gpu_profile_detectors.sh (FETCH_SIZE,
WRITE_SIZE; separate runs as the MI355X guide prescribes): mean KB per dispatch and hbm bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(gfx950: FETCH_SIZE tallies 64 B per 128-B request on wide coalesced reads; these kernels' gathers are narrower, so 2x is an UPPER
bound for them -- the figures say "a few hundred KB per launch", which is all they are used for).
Usage: python scripts/pmc_detectors.py fetch.db write.db > profiles/<tag>_pmc.txt"""
import sqlite3, statistics, sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, value, start from counters_collection where counter_name=? order by start", (counter,)).fetchall()
    by = {}
    for k, v, s in rows:
        by.setdefault(k.replace("(anonymous namespace)::", "").split("(")[0], []).append(v)
    return by


F, W = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
print("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over scripts/gpu_bench_detectors.py, MI355X")
print("# mean per dispatch over all dispatches of the run (2D: 1440 / 3600 beams; 3D: 28.8 k / 57.6 k points mixed)")
print(f"{'kernel':28s} {'calls':>6s} {'FETCH_KB':>10s} {'WRITE_KB':>10s} {'hbm_KB<=':>10s}")
for k in sorted(set(F) | set(W)):
    if "rocclr" in k:
        continue
    f = statistics.mean(F.get(k, [0])); w = statistics.mean(W.get(k, [0]))
    print(f"{k[:28]:28s} {len(F.get(k, [])):6d} {f:10.1f} {w:10.1f} {2 * f + w:10.1f}")
