"""Summarise the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X guide
prescribes) into profiles/<tag>_pmc.txt and profiles/pmc_downdate.json (read by bench.py for
roofline.traffic).  HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE
reports half the bytes of a wide (16 B/lane) coalesced read (MI355X_MICROARCH.md, HBM section);
WRITE_SIZE is taken as reported -- the 35,684,352-byte hipMemset of P reads back as exactly 34848 KB,
which calibrates it.  Usage: python scripts/pmc_summary.py fetch.db write.db tag [last_n]"""
import json, sqlite3, sys, statistics
fdb, wdb, tag = sys.argv[1], sys.argv[2], sys.argv[3]
last = int(sys.argv[4]) if len(sys.argv) > 4 else 50
outdir = sys.argv[5] if len(sys.argv) > 5 else "profiles"
commit = sys.argv[6] if len(sys.argv) > 6 else None
def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, value, start from counters_collection where counter_name=? order by start", (counter,)).fetchall()
    by = {}
    for k, v, s in rows: by.setdefault(k, []).append(v)
    return {k: v[-last:] for k, v in by.items()}
F, W = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
lines = [f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace) over bench.py; last {last} dispatches per kernel, C3 (n=2051, m=64), MI355X",
         f"# units: KB as reported; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request on 16-B/lane streams)",
         f"{'kernel':44s} {'FETCH_KB':>10s} {'WRITE_KB':>10s} {'hbm_MB':>8s}"]
out = {}
for k in sorted(set(F) | set(W)):
    f = statistics.mean(F.get(k, [0])); w = statistics.mean(W.get(k, [0]))
    hbm = (2 * f + w) * 1024
    lines.append(f"{k[:44]:44s} {f:10.1f} {w:10.1f} {hbm/1e6:8.2f}")
    # since round 5 an update is ONE launch, k_mid<4, 0> (mid role + the previous scan's downdate role + the next scan's speculative
    # front end): its traffic is the downdate's plus a few MB of gathers and panels
    if "k_mid<4, 0>" in k:
        out = {"kernel": k, "fetch_size_kb": f, "write_size_kb": w, "hbm_bytes_per_launch": hbm, "_commit": commit,
               "note": "2*FETCH_SIZE + WRITE_SIZE, KB->bytes; see profiles/%s_pmc.txt" % tag}
open(f"{outdir}/{tag}_pmc.txt", "w").write("\n".join(lines) + "\n")
json.dump(out, open(f"{outdir}/pmc_downdate.json", "w"), indent=1)
print("\n".join(lines)); print(out)
