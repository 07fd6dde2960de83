import sys, json
rows=[json.loads(l) for l in sys.stdin if l.startswith("{")]
print(rows[-1])
R=[r for r in rows if not r.get("summary")]
st=[r for r in R if r["steady_scans"]>0]
print(len(st), "sessions with a steady tail;", sum(r["steady_scans"] for r in st), "steady scans; worst mu", max(r["worst_mu"] for r in R), "worst cov", max(r["worst_cov"] for r in R))
for r in R:
    if not r["ok"]: print(r)
