"""bench.py's one-line JSON contract (the driver parses it): required keys, types, the roofline / cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "200", "--warmup", "20", "--cpu-literal-steps", "1",
                          "--cpu-structured-steps", "5", "--multi-sessions", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict),
                 ("cpu_baseline", dict)):
        assert isinstance(d[k], t), (k, d[k])
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 200 and d["warmup"] == 20 and d["higher_is_better"] is True
    assert d["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert d["dtype"] == "f64" and d["scaling"] == "weak" and "workload" in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.1 < r["frac"] < 1.0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0
    assert d["value"] > 5000 and d["pose_rmse_vs_oracle_m"] < 1e-5           # the north-star bars
    assert d["multi_session"]["sessions_bit_identical"] is True
