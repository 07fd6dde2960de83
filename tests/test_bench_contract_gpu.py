"""bench.py's one-line JSON contract (the driver parses it): required keys, types, the roofline / cpu_baseline objects.
The first test runs the DRIVER'S EXACT COMMAND (`--gpus 1 --steps 20 --warmup 5`): round 1's cpu_baseline was empty at
exactly that setting, so every object on the line is checked there."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, env=None):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def _check_contract(d, steps, warmup):
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict)):
        assert isinstance(d[k], t), (k, d[k])
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup and d["higher_is_better"] is True
    assert d["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert d["dtype"] == "f64" and d["scaling"] == "weak" and "workload" in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "avg_launch_us_method"):
        assert k in r
    # (round 6: the launch is bound by the mid role's latency chain and says so; of the two rooflines the HBM one is the nearer)
    assert r["bound"].startswith("latency") and r["bound_hbm_or_mfma"] == "hbm" and "NOT met" in r["bytes_note"] and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.05 < r["frac"] < 1.0
    assert r["traffic"] is None or "NOT measured in this run" in r["traffic_source"]
    # honest accounting: frac counts what the lower-triangle algorithm must move; the full-square figure rides along
    assert r["frac_fullsquare"] > 1.5 * r["frac"] and r["bytes_per_launch"] < 0.55 * r["bytes_per_launch_fullsquare"]   # triangle read + written
    # `frac` is measured in this run over the ONE launch an update is (its period in an un-instrumented window); the downdate role's own span
    # inside that launch (device time stamps) and the stand-alone kernel back to back ride along
    assert "ONE launch" in r["avg_launch_us_method"] and "500-update window" in r["avg_launch_us_method"] and r["avg_launch_us"] > 1.0
    assert "frac_back_to_back" not in r and "frac_back_to_back" not in r["mfma"]      # (the stand-alone, Infinity-Cache-resident re-launch figure is gone from the line)
    assert abs(r["frac"] - r["bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9 / r["peak"]) < 1e-9
    assert r["downdate_role_us"] is not None and 2.0 < r["downdate_role_us"] < r["avg_launch_us"] * 1.2
    assert r["frac"] <= r["frac_downdate_role"] * 1.2 < 1.2
    assert r["frac_moved"] is None or 0.05 < r["frac_moved"] < 1.0
    # every rank's record carries its parity figure (here: the one rank), taken after the timed region
    pe = d["ranks"]["max_abs_err_vs_oracle"]
    assert len(pe) == 1 and pe[0] is not None and pe[0] < 1e-9, pe
    assert r["mfma"]["frac"] < r["mfma"]["frac_fullsquare_flop"] and 0.02 < r["mfma"]["frac"] < 1.0
    assert d["value"] > 5000                                                   # the north-star bar is 10 k; 20-step runs are noisy
    # (round 6) how the pipeline fared in the window the roofline's duration comes from: the wrapper-default filter speculated nearly every scan
    pl = d["pipeline"]
    assert pl["window_updates"] == 500 and pl["speculative_records_proved"] > 450 and pl["scans_beside_a_pending_downdate"] > 450
    assert 0 <= pl["of_those_computing_its_correction_themselves"] <= pl["scans_beside_a_pending_downdate"]


def test_driver_command_steps20_warmup5_has_every_object():
    d = _run("--gpus", "1", "--steps", "20", "--warmup", "5")
    _check_contract(d, 20, 5)
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c
    assert c["kind"] == "port" and c["cores"] == 1
    # the literal (reference-formulation) update costs seconds at n = 2051 on one core: anything far outside is a bug
    assert c["value"] is not None and 0.02 < c["value"] < 20.0, c
    assert c["sample"].startswith("3 steady-state updates") and c["literal_s_per_update"] > 0.05
    assert 1.0 < c["structured_value"] < 5000.0
    assert d["pose_rmse_vs_oracle_m"] < 1e-5                                   # north-star parity bar
    lat = d["latency_us"]
    for leg in ("device_chain", "host_sync"):
        assert lat[leg]["n"] == 1000 and 5.0 < lat[leg]["median"] <= lat[leg]["p99"] < 5000.0
    p5 = d["with_5_predicts_per_scan"]
    assert p5["predicts_per_scan"] == 5 and 1000 < p5["value"] < d["value"] * 1.2
    assert d["kernel_us"]["mid"] > 1.0 and d["kernel_us"]["downdate"] is None   # (the downdate is a role of k_mid's launch: no launch of its own in the steady state)
    ex = lat["exclusive_handle"]
    assert 5.0 < ex["host_sync_median"] < 5000.0 and 5.0 < ex["host_sync_idle_device_median"] < 5000.0
    assert c["structured_all_cores"] >= 1 and c["structured_all_cores_value"] > 1.0
    assert "fixed_capacity" in d and "error" not in d["fixed_capacity"] and d["fixed_capacity"]["value"] > 5000
    assert d["config"]["max_landmarks"] >= 1024 and d["not_full"]["value"] > 0.8 * d["fixed_capacity"]["value"]     # (a growing filter takes the one-launch form, too)
    assert 0.05 < d["kernel_us"]["odometry_message_plus_get_pose_host_us"] < 50.0     # no launch: host pose mirror
    nf = d["not_full"]
    assert "error" not in nf, nf
    assert nf["max_landmarks"] == 2048 and nf["n"] == 2051 and nf["value"] > 5000 and nf["with_pose_readback"]["value"] > 3000
    assert nf["kernel_us"]["augment"] is None          # k_augment has no launch of its own: the last downdate workgroup of k_dd_front to finish appends (no waiting workgroup either)
    assert nf["exclusive_handle"]["value"] > 5000
    assert d["multi_session"]["sessions_bit_identical"] is True
    det = d["detectors"]
    assert "error" not in det, det
    l2, c3 = det["laser_2d"], det["cloud_3d"]
    assert l2["beams"] == 3600 and l2["reflectors"] > 5 and 3.0 < l2["call_us"]["median"] <= l2["call_us"]["p99"] < 2000.0
    assert c3["points"] == 28800 and c3["reflectors"] > 5 and 10.0 < c3["call_us"]["median"] <= c3["call_us"]["p99"] < 5000.0
    assert l2["identical_to_oracle"] is True and c3["identical_to_oracle"] is True
    assert l2["cpu_oracle_us"]["median"] > 5.0 and c3["cpu_oracle_us"]["median"] > 20.0 * c3["call_us"]["median"]
    assert abs(c3["gb_per_s"] - c3["input_bytes"] / (c3["call_us"]["median"] * 1e-6) / 1e9) < 1e-9
    sec = d["secondary"]
    assert set(sec) == {"C2", "C4"} and all("error" not in v for v in sec.values()), sec
    assert sec["C2"]["n"] == 259 and sec["C2"]["m"] == 32 and sec["C2"]["value"] > 5000
    assert sec["C4"]["n"] == 1027 and sec["C4"]["odom_model"] == "omni" and sec["C4"]["value"] > 5000
    assert sec["C4"]["detector_pipeline"]["value"] > 100 and sec["C4"]["detector_pipeline"]["final_n"] == 1027


def test_bench_light_run_steps200():
    d = _run("--steps", "200", "--warmup", "20", "--no-cpu-baseline", "--multi-sessions", "0", "--secondary", "", "--latency-steps", "0")
    _check_contract(d, 200, 20)
    assert "cpu_baseline" not in d and "secondary" not in d and "latency_us" not in d
    assert "cpu_oracle_us" not in d["detectors"]["cloud_3d"]                    # the oracle is only timed in the CPU-baseline leg


def test_rccl_path_at_world_size_one():
    """The collectives of the session-per-GPU harness on the ONE GPU there is (no 8-GPU node is reachable from the builder): process
    group "nccl" (= RCCL) with device_id, barrier, all_reduce(MAX) of a cuda float64, all_gather of the 64-byte records."""
    d = _run("--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--secondary", "", "--multi-sessions", "0",
             "--latency-steps", "0", "--detector-reps", "0", env={"REKF_BENCH_FORCE_DIST": "1", "MASTER_PORT": "29541"})
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["dist_backend"] == "nccl"
    r = d["ranks"]
    assert r["final_n"] == [2051] and len(r["seeds"]) == 1
    assert r["max_abs_err_vs_oracle"][0] is not None and r["max_abs_err_vs_oracle"][0] < 1e-9
    assert abs(d["value"] - 20 / (d["ms_per_step"] * 1e-3 * 20)) < 1e-6 * d["value"]
