#!/usr/bin/env python
"""bench.py -- EKF updates/s at N=1024 landmarks x 32 observations/scan (BASELINE.json
configs[2], "C3") on MI355X, one filter session per GPU.

A "step" is one steady-state HandleObservationMessage (predict + ReflectorMatch +
EKF update; reference reflector_ekf_slam.cc:229-309) on a filter whose covariance
(n = 2051, 33.6 MB FP64) is resident in HBM.  The map-building warm-up (the augment
path) runs untimed before it.  Inputs per step are the scan's 32 float32 points,
passed by value with the launch; nothing else crosses PCIe in the timed region.

    python bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no torchrun environment re-executes itself under
`python -m torch.distributed.run --nproc-per-node N` (one rank per GPU over RCCL);
launched by the driver under torchrun it uses the ranks it is given.  Ranks run
independent sessions (seed + rank, BASELINE.json configs[4]) -- "replicas only", no
data-path collective: a start barrier, MAX over ranks of the elapsed time and one
all-gather of a 64-byte result record per rank (reflector_ekf_slam_amd/dist.py).
Rank 0 prints ONE JSON line.

Extra objects on that line (rank 0, N = 1 unless noted):
  roofline       the update's ONE launch, k_mid<4, 0>: the scan's mid role (a latency chain), the previous scan's downdate P -= K (H P) as a
                 role beside it (the HBM / MFMA work) and the next scan's speculative front end.  `frac` / `achieved` / `mfma.frac` are over
                 the launch's average duration measured in this run (the update period of an un-instrumented window); P is stored as its LOWER
                 TRIANGLE, so ("scale both FLOP and BYTES by the executed tile fraction") they count the bytes that algorithm
                 must move -- the triangle read and written, the two panels -- and the MFMA FLOP actually executed; the full-square SURVEY
                 figure (16 n^2 + 8 n (3+m)) is kept as `frac_fullsquare`, the PMC-measured bytes as `frac_moved`.  `bound` says what
                 really binds the launch (the latency chain); `traffic` and `mfma.counter_busy_cycles` are NOT measured in this run: they are
                 read from the committed rocprofv3 PMC summaries (profiles/MANIFEST.json names the commit and the sources they describe).
  not_full       the same steady state on a filter created with a FIXED capacity of 2 L (auto_grow off): like the headline's wrapper-default
                 filter it can still grow, so the host learns each scan's n from that scan's k_mid (rekf.h, PENDING WORK) and runs the same
                 one-launch form.  `fixed_capacity` = max_landmarks = L, auto_grow off: the configuration rounds 1-5 measured the headline on.
  latency_us     median / p99 of one update: hipEvent pair around each whole chain
                 (device) and host wall time of HandleObservationMessage + GetPose.
  with_5_predicts_per_scan   the same scans with five HandleOdometryMessage
                 predicts in front of each (the reference's 50 Hz odometry : 10 Hz scans).
  cpu_baseline   the CPU oracle (oracle/ekf_oracle.c, "port") in LITERAL mode -- the
                 dense operation sequence the reference's Eigen expressions execute --
                 timed on this box's host, 1 thread, on its own bounded sample of
                 steady-state scans (independent of --steps), from the GPU's state.
  secondary      the same measurement (rate + per-kernel us) at BASELINE.json
                 configs[1] (C2) and configs[3] (C4: omni odometry; also with the 3D
                 detector in front of the filter) -- parity-test configs, not `value`.
  multi_session  aggregate updates/s of 4 independent sessions sharing GPU 0.
  detectors      reflector_detect's two front ends at the C ABI, measured in this run: HandleLaserScan on a 3600-beam
                 scan and HandlePointCloud on a 28.8 k-point sweep -- call us (median / p99, host buffers in, centres
                 out: PCIe inclusive), input bytes / call time in GB/s, and the CPU oracle's time for the same inputs
                 (part of the CPU-baseline leg: skipped with --no-cpu-baseline).
  ranks          per-rank updates/s: min / median / max, and every rank's parity figure -- max |mu - oracle| over
                 `--rank-parity-steps` updates run AFTER the timed region, oracle started from the rank's own state.
"""
from __future__ import annotations

import os

# The multi_session leg runs 4 sessions next to the main handle: five HIP streams.  ROCm maps streams onto 4 hardware
# queues per process by default (two sessions would share one and serialise); must be set before the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# the CPU baseline's all-core leg (oracle/ekf_oracle.c, one persistent OpenMP team per update): threads pinned to consecutive cores --
# one socket -- so that each thread's share of P stays in its own cache.  Read by libgomp when it starts, i.e. before anything loads it.
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")

import argparse
import dataclasses
import json
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_MFMA_PEAK_TF = 78.6       # MI355X datasheet FP64 matrix (v_mfma_f64_16x16x4_f64: 32 FLOP/clk/SIMD)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", default="C3", choices=["C2", "C3", "C4", "T0"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--multi-sessions", type=int, default=4,
                    help="also report the aggregate rate of this many independent sessions sharing GPU 0 (0: skip)")
    ap.add_argument("--cpu-literal-steps", type=int, default=3)
    ap.add_argument("--cpu-structured-steps", type=int, default=60)
    ap.add_argument("--latency-steps", type=int, default=1000, help="updates in each per-update latency pass (0: skip)")
    ap.add_argument("--instr-steps", type=int, default=300, help="updates in the per-kernel hipEvent pass")
    ap.add_argument("--secondary", default="C2,C4", help="comma list of other BASELINE configs to also measure (rank 0, N=1)")
    ap.add_argument("--secondary-steps", type=int, default=1000)
    ap.add_argument("--rank-parity-steps", type=int, default=20,
                    help="updates every rank replays against the CPU oracle AFTER the timed region (its parity figure in `ranks`; 0: skip)")
    ap.add_argument("--detector-reps", type=int, default=200, help="calls per detector in the `detectors` leg (0: skip)")
    ap.add_argument("--fixed-capacity", action="store_true",
                    help="build the headline filter with max_landmarks = L, auto_grow off (rounds 1-5's configuration) instead of the wrapper's "
                         "defaults: for A/B and counter passes; the line's config says so")
    ap.add_argument("--timed-only", action="store_true",
                    help="map build, warm-up and the timed steps only (no instrumented legs): what scripts/gpu_profile_round.sh "
                         "runs under rocprofv3, so that the LAST dispatches of every kernel are the timed region's")
    return ap.parse_args(argv)


def config_by_name(name):
    from reflector_ekf_slam_amd import synth
    if name == "T0":       # tiny: exercises this file's plumbing in the CPU tests (gloo, stub filter)
        return synth.SessionConfig("T0_N12_obs6", 12, 6, synth.DIFF, seed=77, speed=1.0, row_spacing=6.0)
    return getattr(synth, name)


def metric_text(cfg):
    return f"EKF updates/s at N={cfg.n_landmarks} landmarks, {cfg.obs_per_scan} obs/scan; pose RMSE vs reference"


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` outside torchrun: become N ranks (one per GPU) ourselves."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execvpe(cmd[0], cmd, env)


# ---------------------------------------------------------------------------------------------------
# the part every rank runs (also driven by tests/test_host_cpu.py with a CPU stub filter over gloo)
# ---------------------------------------------------------------------------------------------------
def gpu_filter_factory(cfg, sess, device, capacity=None):
    """The filter every leg runs on.  Default: the wrapper's OWN defaults -- ReflectorEKFSLAM(options): auto_grow on, initial capacity 1024
    reflectors, doubling on demand like the reference's ever-growing state (cc:311-364) -- which is what the headline is measured on since
    round 6.  `capacity` = a FIXED capacity (auto_grow off): the `not_full` leg (2 L) and the `fixed_capacity` leg (L: round 5's headline form)."""
    from reflector_ekf_slam_amd import ReflectorEKFSLAM
    from reflector_ekf_slam_amd import session as S
    if capacity is None:
        return ReflectorEKFSLAM(S.options_for(sess), device=device)
    return ReflectorEKFSLAM(S.options_for(sess), max_landmarks=capacity, device=device, auto_grow=False)


def build_session(cfg_name, rank, world):
    from reflector_ekf_slam_amd import synth
    base = config_by_name(cfg_name)
    cfg = dataclasses.replace(base, seed=base.seed + (10 + rank if world > 1 else 0),
                              name=base.name + (f"_rank{rank}" if world > 1 else ""))
    return base, cfg, synth.make_session(cfg)


def timed_region(ekf, scans, warmup, steps, dist_mod, device_sync):
    """W untimed warm-up steps, then exactly K steps between barrier + sync on both sides.
    Returns this rank's elapsed seconds and the number of scans consumed."""
    from reflector_ekf_slam_amd import dist as D
    it = iter(scans)
    for _ in range(warmup):
        t, ob = next(it)
        ekf.handle_observation(t, ob)
    D.barrier(dist_mod)
    device_sync()
    ekf.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        t, ob = next(it)
        ekf.handle_observation(t, ob)
    ekf.sync()
    device_sync()
    elapsed = time.perf_counter() - t0
    D.barrier(dist_mod)
    return elapsed, warmup + steps


def rank_parity(cfg, sess, ekf, scans):
    """The rank's parity figure (every rank's record carries its max-abs pose error): the CPU oracle
    (structured mode) is started from THIS rank's state after the timed region and both replay `scans`; returns
    max |mu_gpu - mu_oracle| over all of mu after the last one and whether every association list was identical.
    Not timed, not part of `value`; the oracle is only the checker here, as in the cpu_baseline leg."""
    from oracle.binding import OracleEKF
    st = ekf.GetState()
    o = OracleEKF(cfg.odom_model, sess.init_time, sess.init_pose, cfg.sigma_v ** 2, cfg.sigma_w ** 2, cfg.sigma_obs ** 2)
    vt = sess.odom[np.nonzero(sess.ev_type == 0)[0][-1]]
    o.set_state(st.time, st.mu, st.sigma, vt)
    same = True
    for t, ob in scans:
        ekf.handle_observation(t, ob)
        o.handle_observation(t, ob)
        mg = ekf.last_match()
        gs, gn = (mg[0], mg[2]) if isinstance(mg, tuple) else (mg.state_obs_match_ids, mg.new_ids)
        sp, mp, nw = o.last_match()
        same = same and bool(np.array_equal(np.asarray(gs).reshape(-1, 2), np.asarray(sp).reshape(-1, 2)) and np.array_equal(gn, nw))
    return float(np.abs(ekf.mu() - o.mu()).max()), same


def run_rank(args, dist_mod, rank, local_rank, world, make_filter=gpu_filter_factory, device_sync=lambda: None,
             full=True):
    """One rank's share of the bench.  `full` = also the rank-0 instrumented legs that need the HIP handle."""
    from reflector_ekf_slam_amd import dist as D
    from reflector_ekf_slam_amd import session as S
    from reflector_ekf_slam_amd import synth

    base, cfg, sess = build_session(args.config, rank, world)
    L = cfg.n_landmarks
    n_expect = 3 + 2 * L
    ekf = make_filter(cfg, sess, local_rank, cfg.n_landmarks) if getattr(args, "fixed_capacity", False) else make_filter(cfg, sess, local_rank)

    # ---- untimed: build the map (augment path) ---------------------------------
    t0 = time.time()
    S.replay(sess, ekf)
    ekf.sync()
    n = ekf.n
    map_build_s = time.time() - t0
    if n != n_expect:
        raise SystemExit(f"warm-up ended with n={n}, expected {n_expect}")

    n_extra = 16 + 3 * max(args.latency_steps, 0) + 700 + max(args.instr_steps, 0) + 2 * max(args.steps, 200) + max(args.rank_parity_steps, 0) + 700
    steady = synth.steady_state_scans(sess, args.warmup + args.steps + n_extra)
    m = 2 * steady[0][1].shape[0]

    elapsed, used = timed_region(ekf, steady, args.warmup, args.steps, dist_mod, device_sync)
    elapsed_max = D.max_over_ranks(dist_mod, elapsed)
    mm = ekf.last_match()
    sp = mm[0] if isinstance(mm, tuple) else mm.state_obs_match_ids
    nw = mm[2] if isinstance(mm, tuple) else mm.new_ids
    assert len(nw) == 0 and len(sp) == m // 2, "not steady state"
    assert ekf.n == n_expect
    # every rank, after the timed region: its own parity figure (the gloo CPU test drives this function with an oracle stub: nothing to compare)
    err, assoc_ok = float("nan"), True
    if args.rank_parity_steps > 0 and hasattr(ekf, "GetState"):
        err, assoc_ok = rank_parity(cfg, sess, ekf, steady[used:used + args.rank_parity_steps])
        used += args.rank_parity_steps
    mu = ekf.mu()
    recs = D.gather_records(dist_mod, dict(steps=args.steps, elapsed_s=elapsed, final_n=ekf.n, pose_x=mu[0], pose_y=mu[1],
                                           pose_theta=mu[2], max_abs_err=err if assoc_ok else float("inf"), seed=cfg.seed))
    if rank != 0:
        return None
    rates = sorted(r["steps"] / r["elapsed_s"] for r in recs)
    out = {
        "metric": metric_text(base),
        "value": D.aggregate_updates_per_s(recs, elapsed_max),
        "unit": "updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed_max / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{base.name}: synthetic 2D session, L={L} landmarks (n={n}), "
                               f"{m // 2} matched observations/scan (m={m}), "
                               f"{'diff-drive' if cfg.odom_model == synth.DIFF else 'omni'} odometry, "
                               "steady-state HandleObservationMessage; filter built with " +
                               ("max_landmarks = L, auto_grow off (--fixed-capacity)" if getattr(args, "fixed_capacity", False) else
                                "the wrapper's defaults (ReflectorEKFSLAM(options): auto_grow, initial capacity 1024)"),
                   "max_landmarks": int(getattr(ekf, "max_landmarks", 0) or 0),
                   "sessions": world, "parallelism": "replicas: one independent session per GPU",
                   "map_build_s": round(map_build_s, 3)},
        "dist_backend": (dist_mod.get_backend() if dist_mod is not None else None),
        "ranks": {"updates_per_s_min": rates[0], "updates_per_s_median": float(np.median(rates)),
                  "updates_per_s_max": rates[-1], "final_n": [int(r["final_n"]) for r in recs],
                  "seeds": [int(r["seed"]) for r in recs],
                  "max_abs_err_vs_oracle": [None if r["max_abs_err"] != r["max_abs_err"] else float(r["max_abs_err"]) for r in recs],
                  "parity_note": f"per rank: max |mu - CPU oracle| after {args.rank_parity_steps} updates replayed by both from the rank's own "
                                 "state behind the timed region (inf: an association list differed; null: not run)"},
    }
    if not full or args.timed_only:
        return out
    rest = steady[used:]
    out.update(instrumented_legs(args, base, cfg, sess, ekf, rest, n, m, local_rank, world, chain_us=1e6 * elapsed / args.steps))
    return out


# ---------------------------------------------------------------------------------------------------
# rank-0 legs that need the HIP handle
# ---------------------------------------------------------------------------------------------------
def latency_legs(ekf, scans, steps):
    """Per-update latency, two ways, `steps` updates each (never part of `value`)."""
    if steps <= 0:
        return None, 0
    # (a) device: one hipEvent pair around each whole chain on the handle's stream, launches back to back
    ekf.sync()
    ekf.profile_reset()
    ekf.profile(True, only=["update"])
    for t, ob in scans[:steps]:
        ekf.handle_observation(t, ob)
    ekf.profile(False)
    dev = ekf.profile_update_samples()
    ekf.profile_reset()
    # (b) host: what the reference's caller sees -- HandleObservationMessage, then the pose (GetState's fast path)
    host = np.zeros(steps)
    for k, (t, ob) in enumerate(scans[steps:2 * steps]):
        t0 = time.perf_counter()
        ekf.handle_observation(t, ob)
        ekf.pose()
        host[k] = 1e6 * (time.perf_counter() - t0)

    # (c) the same from an IDLE device -- a robot's scans arrive milliseconds apart, long after the previous update's downdate is
    # through: what it waits for between handing a scan over and having its pose
    n_idle = min(steps, 300)
    idle = np.zeros(n_idle)
    for k, (t, ob) in enumerate(scans[2 * steps:2 * steps + n_idle]):
        ekf.sync()
        t0 = time.perf_counter()
        ekf.handle_observation(t, ob)
        ekf.pose()
        idle[k] = 1e6 * (time.perf_counter() - t0)

    def q(a):
        return {"median": float(np.median(a)), "p99": float(np.percentile(a, 99)), "mean": float(a.mean()), "n": int(a.size)}
    return {"device_chain": dict(q(dev), method="hipEvent pair around each whole update chain on the handle's stream, "
                                                  "launches back to back (includes ~4-5 us of event bracket)"),
            "host_sync": dict(q(host), method="host wall: HandleObservationMessage + rekf_get_pose per update, back to back (includes what is left "
                                              "of the previous update's downdate; pose, 3x3 block, n and flags arrive as tagged slots in pinned "
                                              "host memory from k_mid, the host polls them)"),
            "host_sync_idle_device": dict(q(idle), method="the same with the device idle when the scan is handed over (rekf_sync in front, untimed): "
                                                          "front end + k_mid + the PCIe write")}, 2 * steps + n_idle


def predict_leg(ekf, cfg, scans, steps, per_scan=5):
    """Updates/s when every scan is preceded by `per_scan` HandleOdometryMessage predicts (reference
    src/ros_node.cc:627-660: 50 Hz odometry against 10 Hz scans)."""
    rng = np.random.Generator(np.random.PCG64(cfg.seed + 4242))
    odo = rng.normal(0.0, 1.0, size=(steps, per_scan, 2)) * np.array([cfg.sigma_v, cfg.sigma_w])
    t_prev = ekf.GetLatestTime()
    ekf.sync()
    t0 = time.perf_counter()
    for k, (t, ob) in enumerate(scans[:steps]):
        for j in range(per_scan):
            ekf.handle_odometry(t_prev + (t - t_prev) * (j + 1) / (per_scan + 1), odo[k, j, 0], 0.0, odo[k, j, 1])
        ekf.handle_observation(t, ob)
        t_prev = t
    ekf.sync()
    dt = time.perf_counter() - t0
    ekf.handle_odometry(t_prev, 0.0, 0.0, 0.0)        # dt = 0: leaves the state alone, parks the stored velocity
    return {"value": steps / dt, "unit": "updates/s", "us_per_scan": 1e6 * dt / steps, "predicts_per_scan": per_scan,
            "steps": steps}, steps


def per_kernel_leg(ekf, scans, steps):
    ekf.profile_reset()
    ekf.profile(True, only=["front", "mid", "downdate", "augment", "empty"])
    for t, ob in scans[:steps]:
        ekf.handle_observation(t, ob)
    ekf.profile(False)
    prof = ekf.profile_read()
    ekf.profile_reset()
    # (a kernel that ran for only a few of the updates has no representative figure: with the lazy downdate the front end runs inside
    # k_dd_front -- bracketed as "downdate" -- and k_front_mb only behind a read-back)
    return {k: (round(v[0] / v[1], 3) if v[1] >= max(2, steps // 4) else None) for k, v in prof.items() if k != "update"}


def instrumented_legs(args, base, cfg, sess, ekf, rest, n, m, device, world, chain_us=None):
    out = {}
    pos = 0
    lat, used = latency_legs(ekf, rest[pos:], args.latency_steps)
    pos += used
    if lat is not None:
        out["latency_us"] = lat
        # ... and on an EXCLUSIVE handle (rekf_set_exclusive: the caller promises the GPU is this handle's alone -- a bench process is):
        # the scan's front end then runs inside the scan's launch, the update's workgroups waiting for it in there
        try:
            ekf.set_exclusive(True)
            k_ex = min(args.latency_steps, 300)
            hx, ix = np.zeros(k_ex), np.zeros(k_ex)
            for k, (t, ob) in enumerate(rest[pos:pos + k_ex]):
                t0 = time.perf_counter(); ekf.handle_observation(t, ob); ekf.pose(); hx[k] = 1e6 * (time.perf_counter() - t0)
            for k, (t, ob) in enumerate(rest[pos + k_ex:pos + 2 * k_ex]):
                ekf.sync()
                t0 = time.perf_counter(); ekf.handle_observation(t, ob); ekf.pose(); ix[k] = 1e6 * (time.perf_counter() - t0)
            pos += 2 * k_ex
            out["latency_us"]["exclusive_handle"] = {"host_sync_median": float(np.median(hx)), "host_sync_p99": float(np.percentile(hx, 99)),
                                                     "host_sync_idle_device_median": float(np.median(ix)), "host_sync_idle_device_p99": float(np.percentile(ix, 99)), "n": k_ex}
        finally:
            ekf.set_exclusive(False)
    k5 = max(args.steps, 200)
    out["with_5_predicts_per_scan"], used = predict_leg(ekf, cfg, rest[pos:], k5)
    pos += used
    # odometry path alone: HandleOdometryMessage + GetPose at odometry rate (src/ros_node.cc:627-660) -- no launch, host mirror
    ekf.pose()
    tp = ekf.GetLatestTime()
    t0 = time.perf_counter()
    for j in range(2000):
        ekf.handle_odometry(tp + 1e-6 * (j + 1), 0.01, 0.0, 0.01)
        ekf.pose()
    odo_us = 1e6 * (time.perf_counter() - t0) / 2000
    ekf.handle_odometry(tp + 1e-6 * 2001, 0.0, 0.0, 0.0)

    ms_result = multi_session(args, cfg, sess, device) if (world == 1 and args.multi_sessions > 1) else None

    state_for_cpu = None
    if not args.no_cpu_baseline:
        state_for_cpu = ekf.GetState()
    kernel_us = per_kernel_leg(ekf, rest[pos:], args.instr_steps)
    kernel_us["odometry_message_plus_get_pose_host_us"] = round(odo_us, 3)     # no kernel: Predict runs on the host's pose mirror
    pos += args.instr_steps
    # (the chain's us per update from a window of its own, 500 updates: the driver's 20-step timed region carries its pipeline fill and
    # its closing synchronisations 25 times harder, which is not kernel time)
    def pipeline_counters():
        import ctypes as _Cc
        c = (_Cc.c_longlong * 32)()
        return [int(x) for x in c] if ekf._L.rekf_debug_counters(ekf._h, c) == 0 else None
    if len(rest) - pos >= 500:
        ekf.sync()
        c0 = pipeline_counters()
        t0c = time.perf_counter()
        for t, ob in rest[pos:pos + 500]:
            ekf.handle_observation(t, ob)
        ekf.sync()
        chain_us = 1e6 * (time.perf_counter() - t0c) / 500
        c1 = pipeline_counters()
        pos += 500
        if c0 and c1:
            dlt = [b - a for a, b in zip(c0, c1)]
            out["pipeline"] = {"window_updates": 500,
                               "speculative_records_proved": dlt[20], "of_those_with_rematched_observations": dlt[21],
                               "scans_beside_a_pending_downdate": dlt[22], "of_those_computing_its_correction_themselves": dlt[23],
                               "grid_matched_scans": dlt[18], "of_those_by_the_full_sweep": dlt[19], "grid_builds_so_far": c1[17],
                               "note": "rekf_debug_counters over the un-instrumented 500-update window the roofline's avg_launch_us comes from: how often "
                                       "the speculative match record stood as proved, how often the write-ahead panel served (the rest computed the pending "
                                       "scan's correction themselves: +6 us on that scan)"}
    # ---- the roofline kernel.  Since round 5 a steady-state update is ONE launch, k_mid<4, 0>: the scan's mid role (gather + 64 x 64
    # inverse + gain: a latency chain), the PREVIOUS scan's rank-m downdate P -= K (H P) as a role beside it (the HBM work: the stored lower
    # triangle read from one P buffer and written to the other, tiles from a queue) and the NEXT scan's speculative front end.  Its average
    # duration is measured in this run as the update period of an un-instrumented window (one launch per update on the handle's stream, back
    # to back: period = launch + launch boundary, so the figure is an upper bound of the kernel time; rocprofv3 --kernel-trace of the same
    # loop is committed under profiles/).  The downdate ROLE's own span inside that launch comes from two device time stamps (its first
    # workgroup's start, its last workgroup's end).  The old stand-alone kernel (k_downdate2<64>: same body, whole GPU) is re-timed back to
    # back between one hipEvent pair for comparison; the filter state is meaningless afterwards, nothing below uses this handle again.
    import ctypes as _C
    dd_role_us = None
    try:
        cnt = (_C.c_longlong * 32)()
        ekf.sync()
        for t, ob in rest[pos:pos + 8]:
            ekf.handle_observation(t, ob)
        pos += 8
        if ekf._L.rekf_debug_counters(ekf._h, cnt) == 0 and cnt[26] > 0 and cnt[27] > cnt[26]:
            dd_role_us = (cnt[27] - cnt[26]) * 0.01
    except Exception:
        pass

    bytes_full = 16.0 * n * n + 8.0 * n * (3 + m)         # SURVEY.md 8(d) BYTES_alg(n, m): every element of P read and written once
    # what the EXECUTED algorithm must move (SURVEY 8(d) for a lower-triangular builder): P is STORED as its lower triangle
    # (round 3), so the triangle is read and written once, the panels once
    bytes_exec = 2.0 * 8.0 * (n * (n + 1) / 2.0) + 8.0 * n * (3 + m)
    T = -(-n // 64)
    tiles_exec = T * (T + 1) // 2
    flop_exec = tiles_exec * 2.0 * 64 * 64 * 64            # MFMA FLOP the downdate role issues (16x16x4 tiles, KC = 64)
    flop_k7 = 2.0 * n * n * m                              # the reference's full-square count
    t_frac = chain_us
    achieved = bytes_exec / (t_frac * 1e-6) / 1e9
    rocprof_us, traffic, traffic_src, rocprof_src = None, None, None, None
    try:
        avg = json.load(open(os.path.join(ROOT, "profiles", "kernel_avg_us.json")))
        rocprof_us = next((v for k, v in avg.items() if "k_mid<4, 0>" in k), None)
        rocprof_src = "profiles/kernel_avg_us.json (committed rocprofv3 --kernel-trace summary of bench.py's timed loop; NOT measured in this run)"
    except Exception:
        pass
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_downdate.json")))
        traffic = pj.get("hbm_bytes_per_launch")
        traffic_src = ("profiles/pmc_downdate.json: (2*FETCH_SIZE + WRITE_SIZE) of two separate rocprofv3 --pmc passes over k_mid<4, 0>, "
                       "committed summary; NOT measured in this run")
    except Exception:
        pass
    mfma_busy, mfma_src = None, None
    try:
        mj = json.load(open(os.path.join(ROOT, "profiles", "pmc_mfma.json")))
        mfma_busy = mj.get("sq_valu_mfma_busy_cycles_median")
        mfma_src = "profiles/pmc_mfma.json (committed rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES summary of k_mid<4, 0>; NOT measured in this run)"
    except Exception:
        pass
    moved = (traffic / (t_frac * 1e-6) / 1e9) if traffic else None
    out["roofline"] = {
        "kernel": "k_mid<4, 0>: ONE launch per update -- the scan's mid role (gather, 64 x 64 inverse, gain: the latency chain that sets the "
                  "launch's length), the PREVIOUS scan's downdate P -= K (H P) as a role on the CUs beside it (FP64 MFMA 16x16x4 tiles of the "
                  "stored lower triangle, panels by LDS DMA, one P buffer read, the other written) and the NEXT scan's speculative front end",
        "bound": "latency (the mid role's dependent chain: match record -> sub-block gather -> S -> 64 x 64 inverse -> gain; the HBM work -- the "
                 "downdate role -- hides under it)",
        "bound_hbm_or_mfma": "hbm",
        "bytes_note": "achieved / frac = bytes the executed lower-triangle algorithm must move per launch: 2 * 8 n(n+1)/2 (triangle read and written) "
                      "+ 8 n (3+m) panels, over the launch's average duration measured in this run (avg_launch_us: the update period of an "
                      "un-instrumented window, one launch per update -- an upper bound of the kernel time).  The launch is NOT HBM-bound: the "
                      "downdate has left the update's critical path and fills the CUs the latency chain leaves free, so frac (of the 8 TB/s HBM peak) "
                      "and mfma.frac (of the FP64 MFMA peak) say how far BOTH rooflines are from binding; north_star's '>= 30 % MFMA utilisation on the "
                      "covariance GEMM' is NOT met in this launch (see mfma.frac / mfma.frac_by_counter).  "
                      "frac_downdate_role = the same bytes over the role's own span inside the launch (device time stamps); "
                      "frac_fullsquare = SURVEY 8(d)'s 16 n^2 + 8 n (3+m) over avg_launch_us; frac_moved = HBM bytes by PMC counters (committed summary)",
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "frac_fullsquare": bytes_full / (t_frac * 1e-6) / 1e9 / HBM_PEAK_GBS,
        "frac_moved": (moved / HBM_PEAK_GBS) if moved else None,
        "downdate_role_us": dd_role_us,
        "frac_downdate_role": (bytes_exec / (dd_role_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if dd_role_us else None,
        "frac_inchain_rocprof": (bytes_exec / (rocprof_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if rocprof_us else None,
        "traffic": traffic, "traffic_source": traffic_src,
        "bytes_per_launch": bytes_exec, "bytes_per_launch_fullsquare": bytes_full,
        "avg_launch_us": t_frac,
        "avg_launch_us_method": "measured in this run: the update period of an un-instrumented 500-update window on the handle's stream, ONE launch "
                                "(k_mid<4, 0>) per update, back to back -- launch + launch boundary, an upper bound of the kernel's duration",
        "rocprof_avg_launch_us": rocprof_us, "rocprof_source": rocprof_src,
        "mfma": {"achieved_tflops": flop_exec / (t_frac * 1e-6) / 1e12, "peak_tflops": FP64_MFMA_PEAK_TF,
                 "frac": flop_exec / (t_frac * 1e-6) / 1e12 / FP64_MFMA_PEAK_TF,
                 "frac_downdate_role": (flop_exec / (dd_role_us * 1e-6) / 1e12 / FP64_MFMA_PEAK_TF) if dd_role_us else None,
                 "frac_fullsquare_flop": flop_k7 / (t_frac * 1e-6) / 1e12 / FP64_MFMA_PEAK_TF,
                 "counter_busy_cycles": mfma_busy, "counter_source": mfma_src,
                 "frac_by_counter": (mfma_busy / (1024.0 * t_frac * 1e-6 * 2.4e9)) if mfma_busy else None,
                 "note": f"frac = the downdate role's EXECUTED MFMA FLOP ({tiles_exec} lower-triangle tiles x 2*64*64*64) over avg_launch_us (the mid "
                         "role's MFMAs -- correction, inverse, gain -- are not counted); frac_downdate_role over the role's own span; "
                         "north_star's 30 % target is not met in the shipped launch (the downdate is sized to hide under the latency chain, not to saturate the MFMA pipes); "
                         "frac_fullsquare_flop = the reference's 2 n^2 m over avg_launch_us; frac_by_counter = SQ_VALU_MFMA_BUSY_CYCLES of the whole launch (both roles; "
                         "committed counter pass) over 1024 SIMDs x avg_launch_us at 2.4 GHz"}}
    out["kernel_us"] = kernel_us
    if ms_result is not None:
        out["multi_session"] = ms_result
    if world == 1:
        try:
            out["not_full"] = not_full_leg(args, base, cfg, sess, device, out_value=None)
        except Exception as e:                 # a secondary figure must never take the headline down
            out["not_full"] = {"error": repr(e)}
        try:
            out["fixed_capacity"] = fixed_capacity_leg(args, cfg, sess, device)
        except Exception as e:
            out["fixed_capacity"] = {"error": repr(e)}
    if world == 1 and args.secondary:
        sec = {}
        for name in [s for s in args.secondary.split(",") if s and s != args.config]:
            try:
                sec[name] = secondary_config(args, name, device)
            except Exception as e:             # a secondary figure must never take the headline down
                sec[name] = {"error": repr(e)}
        out["secondary"] = sec
    if world == 1 and args.detector_reps > 0:
        try:
            out["detectors"] = detectors_leg(args, device)
        except Exception as e:                 # a secondary figure must never take the headline down
            out["detectors"] = {"error": repr(e)}
    if not args.no_cpu_baseline:
        out["cpu_baseline"], out["pose_rmse_vs_oracle_m"] = cpu_baseline(args, cfg, sess, state_for_cpu, device)
    return out


def detectors_leg(args, device):
    """reflector_detect's two front ends at the C ABI (laser_reflector_detect.cc:239-306, point_cloud_reflector_detect.cc:31-97):
    per-call latency with host buffers in and centres out (one synchronising call per scan, like the reference's callback:
    PCIe inclusive), the input bytes over that time, and -- in the CPU-baseline leg -- the CPU oracle on the same inputs."""
    from types import SimpleNamespace as NS
    from reflector_ekf_slam_amd import OdometryData, synth
    from reflector_ekf_slam_amd.detect import (LaserReflectorDetect, PointCloudOptions, PointCloudReflectorDetect,
                                               ReflectorDetectOptions)
    reps = args.detector_reps
    s2b = (0.13686, 0.0, 0.0)
    rng = np.random.Generator(np.random.PCG64(7))

    def lat(f, n):
        f(); f()
        ts = np.zeros(n)
        for k in range(n):
            t0 = time.perf_counter()
            f()
            ts[k] = 1e6 * (time.perf_counter() - t0)
        return ts

    def q(a):
        return {"median": float(np.median(a)), "p99": float(np.percentile(a, 99)), "mean": float(a.mean()), "n": int(a.size)}

    out = {"note": "call = one C-ABI call with host buffers in and the reflector centres out (PCIe inclusive); GB/s = input bytes / median call time "
                   "(these paths are latency chains of a few KB to half a MB: the figure says how far from any bandwidth bound they sit)"}
    # ---- 2D: 3600 beams (ranges + intensities in, centres out), odometry present so that the de-skew runs
    lms = synth.make_world(synth.C2, rng)
    pose = (float(lms[:, 0].mean()), float(lms[:, 1].mean()), 0.6)
    scan = NS(**synth.make_laser_scan(lms, pose, 10.0, rng, n_beams=3600))
    g = LaserReflectorDetect(ReflectorDetectOptions(), sensor_to_base_link=s2b, device=device)
    odo = []
    for k in range(30):
        t = 9.5 + 0.02 * k
        odo.append((t, 0.5 * t, 0.0, 0.0, 1.0, 0.5, 0.0, 0.1))
        g.HandleOdometryData(OdometryData(time=t, position=(0.5 * t, 0.0, 0.0), orientation=(1.0, 0.0, 0.0, 0.0),
                                          linear_velocity=(0.5, 0.0, 0.0), angular_velocity=(0.0, 0.0, 0.1)))
    obs2 = g.HandleLaserScan(scan)
    t2 = lat(lambda: g.HandleLaserScan(scan), reps)
    b2 = 2 * 4 * 3600
    out["laser_2d"] = {"beams": 3600, "reflectors": int(obs2.cloud_.shape[0]), "call_us": q(t2), "input_bytes": b2,
                       "gb_per_s": b2 / (np.median(t2) * 1e-6) / 1e9, "kernel": "k_det2d (one launch per scan)"}
    g.close()
    # ---- 3D: 16 rings x 1800 azimuths = 28.8 k XYZI points
    lms = synth.make_world(synth.C4, rng)
    pose = (float(lms[:, 0].mean()), float(lms[:, 1].mean()), 0.3)
    cloud = synth.make_point_cloud(lms, pose, rng, rings=16, n_az=1800)
    g3 = PointCloudReflectorDetect(PointCloudOptions(), max_points=65536, device=device)
    obs3 = g3.HandlePointCloud(1.0, cloud)
    t3 = lat(lambda: g3.HandlePointCloud(1.0, cloud), reps)
    b3 = 16 * int(cloud.shape[0])
    out["cloud_3d"] = {"points": int(cloud.shape[0]), "reflectors": int(obs3.cloud_.shape[0]), "call_us": q(t3), "input_bytes": b3,
                       "gb_per_s": b3 / (np.median(t3) * 1e-6) / 1e9,
                       "kernels": "k3f_front (gate + sort + boxes), k3_knn, k3_cc_min (+ SOR statistics), k3_cc_link, k3f_clusters: five launches per cloud of at "
                                  "most 5120 survivors of the gate; k3_filter_count/write, k3_scatter, k3_boxes, ..., k3_finish_a, k3_clusters (nine) beyond",
                       "clouds_through_short_chain_and_sent_again": list(g3.debug_path_counts())}
    g3.close()
    if not args.no_cpu_baseline:               # the CPU-baseline leg: the oracle (the checker) timed on the same inputs, and compared
        from oracle.binding import OracleDetect2D, oracle_detect3d
        o = OracleDetect2D(sensor_to_base_link=s2b)
        for rec in odo:
            o.handle_odometry(*rec)
        _, co = o.handle_scan(scan)
        c2 = lat(lambda: o.handle_scan(scan), max(reps // 4, 10))
        out["laser_2d"]["cpu_oracle_us"] = q(c2)
        out["laser_2d"]["identical_to_oracle"] = bool(co.shape == obs2.cloud_.shape and np.array_equal(co, obs2.cloud_))
        co3, _, _ = oracle_detect3d(cloud)
        c3 = lat(lambda: oracle_detect3d(cloud), 3)
        out["cloud_3d"]["cpu_oracle_us"] = q(c3)
        out["cloud_3d"]["identical_to_oracle"] = bool(co3.shape == obs3.cloud_.shape and np.array_equal(co3, obs3.cloud_))
        out["cpu_cores_used"] = 1
    return out


def not_full_leg(args, base, cfg, sess, device, out_value=None):
    """The same steady state on a filter whose capacity is a CAP (max_landmarks = 2 L), not the map size: what a deployed
    node runs all session long (src/ros_node.cc:440 constructs once, landmarks keep arriving).  Every scan may append reflectors
    (cc:311-364: the last downdate workgroup of k_dd_front to finish does it), and while the host runs ahead of the device it does not know n."""
    from reflector_ekf_slam_amd import session as S
    from reflector_ekf_slam_amd import synth
    ekf = gpu_filter_factory(cfg, sess, device, capacity=2 * cfg.n_landmarks)
    S.replay(sess, ekf)
    ekf.sync()
    assert ekf.n == 3 + 2 * cfg.n_landmarks
    steps = max(args.steps, 500)
    scans = synth.steady_state_scans(sess, 100 + 2 * steps + 300 + 520, seed_offset=3000)
    elapsed, used = timed_region(ekf, scans, 100, steps, None, lambda: None)
    res = {"value": steps / elapsed, "unit": "updates/s", "us_per_update": 1e6 * elapsed / steps, "steps": steps,
           "max_landmarks": 2 * cfg.n_landmarks, "n": ekf.n,
           "note": "capacity 2 L: k_dd_front (the previous scan's downdate, whose last workgroup to finish appends that scan's new reflectors, + this scan's front end) -> k_mid per update; k_mid publishes pose and the post-augment n"}
    # ... and as the reference's node drives it: the pose read back after every scan (the host then knows n exactly and predicts itself)
    t0 = time.perf_counter()
    for t, ob in scans[used:used + steps]:
        ekf.handle_observation(t, ob)
        ekf.pose()
    dt = time.perf_counter() - t0
    res["with_pose_readback"] = {"value": steps / dt, "unit": "updates/s", "us_per_update": 1e6 * dt / steps}
    res["kernel_us"] = per_kernel_leg(ekf, scans[used + steps:], 300)
    # ... and on an EXCLUSIVE handle: the previous scan's augmentation rides in the next scan's k_mid (its other workgroups wait for it)
    try:
        ekf.set_exclusive(True)
        base_i = used + steps + 300
        ekf.sync()
        t0 = time.perf_counter()
        for t, ob in scans[base_i:base_i + 500]:
            ekf.handle_observation(t, ob)
        ekf.sync()
        dtx = time.perf_counter() - t0
        res["exclusive_handle"] = {"value": 500 / dtx, "unit": "updates/s", "us_per_update": 1e6 * dtx / 500}
    except Exception as e:
        res["exclusive_handle"] = {"error": repr(e)}
    ekf.close()
    return res


def fixed_capacity_leg(args, cfg, sess, device):
    """Rounds 1-5 measured the headline on a filter created with max_landmarks = L, auto_grow off (n == n_max: the state cannot grow, the
    host knows n without asking).  Kept as a leg of its own so that the rounds stay comparable; `value` is the wrapper-default filter."""
    from reflector_ekf_slam_amd import session as S
    from reflector_ekf_slam_amd import synth
    ekf = gpu_filter_factory(cfg, sess, device, capacity=cfg.n_landmarks)
    S.replay(sess, ekf)
    ekf.sync()
    steps = max(args.steps, 500)
    scans = synth.steady_state_scans(sess, 100 + steps, seed_offset=3100)
    elapsed, _ = timed_region(ekf, scans, 100, steps, None, lambda: None)
    res = {"value": steps / elapsed, "unit": "updates/s", "us_per_update": 1e6 * elapsed / steps, "steps": steps,
           "max_landmarks": cfg.n_landmarks, "auto_grow": False, "note": "round 5's headline configuration (n == n_max)"}
    ekf.close()
    return res


def secondary_config(args, name, device):
    """Rate + per-kernel times at another BASELINE.json config (C2: N=128/16 obs, launch-latency-bound;
    C4: N=512, omni odometry, and the 3D detector -> filter pipeline).  Single session, this GPU."""
    from reflector_ekf_slam_amd import session as S
    from reflector_ekf_slam_amd import synth
    base, cfg, sess = build_session(name, 0, 1)
    ekf = gpu_filter_factory(cfg, sess, device)
    t0 = time.time()
    S.replay(sess, ekf)
    ekf.sync()
    build_s = time.time() - t0
    n = ekf.n
    assert n == 3 + 2 * cfg.n_landmarks, n
    steps = args.secondary_steps
    scans = synth.steady_state_scans(sess, 50 + 2 * steps + 300)
    elapsed, used = timed_region(ekf, scans, 50, steps, None, lambda: None)
    m = 2 * scans[0][1].shape[0]
    res = {"metric": metric_text(base), "value": steps / elapsed, "unit": "updates/s", "us_per_update": 1e6 * elapsed / steps,
           "steps": steps, "n": n, "m": m, "odom_model": "diff" if cfg.odom_model == synth.DIFF else "omni",
           "map_build_s": round(build_s, 3)}
    res["with_5_predicts_per_scan"], u2 = predict_leg(ekf, cfg, scans[used:], steps)
    res["kernel_us"] = per_kernel_leg(ekf, scans[used + u2:], 300)
    if name == "C4":
        res["detector_pipeline"] = c4_detector_pipeline(cfg, sess, ekf, device)
    ekf.close()
    return res


def c4_detector_pipeline(cfg, sess, ekf, device, n_clouds=24, reps=8):
    """BASELINE.json configs[3] as the reference runs it: XYZI cloud -> HandlePointCloud (3D detector, GPU)
    -> HandleObservationMessage, per scan.  Clouds are synthesised up front (host, untimed) at the parked pose."""
    from reflector_ekf_slam_amd import synth
    from reflector_ekf_slam_amd.detect import PointCloudOptions, PointCloudReflectorDetect
    rng = np.random.Generator(np.random.PCG64(cfg.seed + 555))
    pose = sess.true_pose[-1]
    clouds = [synth.make_point_cloud(sess.landmarks, pose, rng, **synth.C4_LIDAR) for _ in range(n_clouds)]
    det = PointCloudReflectorDetect(PointCloudOptions(), max_points=65536, device=device)
    t = ekf.GetLatestTime()
    ks = []
    for c in clouds[:4]:                      # warm-up
        t += 0.1
        ob = det.HandlePointCloud(t, c)
        ekf.handle_observation(t, ob.cloud_[:64])
    ekf.sync()
    t0 = time.perf_counter()
    for r in range(reps):
        for c in clouds:
            t += 0.1
            ob = det.HandlePointCloud(t, c)
            ks.append(ob.cloud_.shape[0])
            ekf.handle_observation(t, ob.cloud_[:64])
    ekf.sync()
    dt = time.perf_counter() - t0
    # ... and with the detector's call in its two halves (rdet3d_submit / rdet3d_collect): cloud k + 1 is copied and enqueued while the device
    # is still on cloud k; the filter takes the scans in the same order, one behind.  Same observations (checked against the synchronous pass).
    seq_obs = []
    for c in clouds:
        t += 0.1
        seq_obs.append(det.HandlePointCloud(t, c).cloud_)
    ekf.sync()
    same = True
    t0 = time.perf_counter()
    for r in range(reps):
        t += 0.1
        det.SubmitPointCloud(t, clouds[0])
        for k in range(1, n_clouds + 1):
            if k < n_clouds:
                det.SubmitPointCloud(t + 0.1, clouds[k])
            ob = det.CollectObservation()
            if r == 0 and not np.array_equal(ob.cloud_, seq_obs[k - 1]): same = False
            ekf.handle_observation(ob.time_, ob.cloud_[:64])
            t += 0.1
    ekf.sync()
    dt2 = time.perf_counter() - t0
    det.close()
    return {"value": reps * n_clouds / dt, "unit": "scans/s (3D detect + EKF update)", "us_per_scan": 1e6 * dt / (reps * n_clouds),
            "points_per_cloud": int(clouds[0].shape[0]), "reflectors_per_scan_mean": float(np.mean(ks)),
            "reflectors_per_scan_max": int(np.max(ks)), "final_n": int(ekf.n),
            "overlapped": {"value": reps * n_clouds / dt2, "us_per_scan": 1e6 * dt2 / (reps * n_clouds), "identical_observations": bool(same),
                           "note": "rdet3d_submit / rdet3d_collect: two clouds on their way; the synchronous call above is the reference's callback"}}


def multi_session(args, cfg, sess, device):
    """Secondary figure (never `value`): S independent filter sessions of the same workload sharing ONE GPU, one handle
    = one HIP stream each, fed round-robin by this host thread.  A single session is a latency-bound chain of
    kernels, so sessions interleave on the device: the aggregate rate is what a fleet
    server gets per GPU.  Every session runs the same scans and must end bit-identical."""
    from reflector_ekf_slam_amd import ReflectorEKFSLAM, synth
    from reflector_ekf_slam_amd import session as S
    ns = args.multi_sessions
    steps = min(max(args.steps, 200), 1000)
    scans = synth.steady_state_scans(sess, 100 + steps)
    handles = []
    for _ in range(ns):
        g = ReflectorEKFSLAM(S.options_for(sess), device=device)          # (the wrapper's defaults, like the headline)
        S.replay(sess, g)
        g.sync()
        handles.append(g)
    for t, ob in scans[:100]:
        for g in handles:
            g.handle_observation(t, ob)
    for g in handles:
        g.sync()
    t0 = time.perf_counter()
    for t, ob in scans[100:]:
        for g in handles:
            g.handle_observation(t, ob)
    for g in handles:
        g.sync()
    dt = time.perf_counter() - t0
    ref = handles[0].mu()
    identical = all(bool(np.array_equal(g.mu(), ref)) for g in handles[1:])
    for g in handles:
        g.close()
    return {"sessions": ns, "value": ns * steps / dt, "unit": "updates/s (aggregate, one GPU)", "steps_per_session": steps,
            "us_per_update_per_session": 1e6 * dt / steps, "sessions_bit_identical": identical,
            "note": "independent sessions on separate streams of one GPU, one host thread; not the headline value"}


def host_topology():
    """(usable logical CPUs, physical cores of ONE socket among them, sockets): what this process may run on -- its affinity mask and the
    cgroup's CPU quota -- read from /proc/cpuinfo and /sys/fs/cgroup; falls back to os.cpu_count()."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except Exception:
        allowed = list(range(os.cpu_count() or 1))
    usable = len(allowed)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            usable = max(1, min(usable, int(float(q) / float(per))))
    except Exception:
        pass
    cores, cur = {}, {}
    try:
        for line in open("/proc/cpuinfo"):
            if ":" not in line:
                if "processor" in cur:
                    cores[int(cur["processor"])] = (int(cur.get("physical id", 0)), int(cur.get("core id", cur["processor"])))
                cur = {}
                continue
            k, v = line.split(":", 1)
            cur[k.strip()] = v.strip()
        if "processor" in cur:
            cores[int(cur["processor"])] = (int(cur.get("physical id", 0)), int(cur.get("core id", cur["processor"])))
    except Exception:
        cores = {}
    phys = {cores[c] for c in allowed if c in cores}
    sockets = sorted({p for p, _ in phys}) or [0]
    one = len({c for p, c in phys if p == sockets[0]}) or usable
    return usable, max(1, min(one, usable)), len(sockets)


def cpu_baseline(args, cfg, sess, st, device):
    """Time the CPU oracle on its OWN bounded sample of steady-state scans (independent of --steps).

    The oracle is started from the state the GPU had after the timed region (st); the structured-mode sample doubles
    as the parity sample (pose RMSE against a scratch GPU handle replaying the same scans from the same state)."""
    from oracle.binding import OracleEKF
    from reflector_ekf_slam_amd import ReflectorEKFSLAM, synth
    from reflector_ekf_slam_amd import session as S
    ns, nl = max(args.cpu_structured_steps, 1), max(args.cpu_literal_steps, 0)
    all_scans = synth.steady_state_scans(sess, ns + nl + 4 * 32, seed_offset=2000)
    shift = st.time - sess.ev_time[-1]            # these scans start right after the snapshot's time
    all_scans = [(t + shift, ob) for t, ob in all_scans]
    o = OracleEKF(cfg.odom_model, sess.init_time, sess.init_pose, cfg.sigma_v ** 2, cfg.sigma_w ** 2,
                  cfg.sigma_obs ** 2)
    vt = sess.odom[np.nonzero(sess.ev_type == 0)[0][-1]]
    o.set_state(st.time, st.mu, st.sigma, vt)
    t0 = time.perf_counter()
    poses = []
    for t, ob in all_scans[:ns]:
        o.handle_observation(t, ob)
        poses.append(o.mu()[:3].copy())
    t_struct = (time.perf_counter() - t0) / ns
    # ... and the same algorithm on the host's cores (the fair algorithmic CPU baseline): ONE persistent OpenMP team per update
    # (oracle/ekf_oracle.c update_structured_team: every thread owns a fixed range of P's columns -- its share stays in its cache from
    # update to update -- and the phases meet at four barriers), threads = the physical cores of one socket this process may use, pinned
    # (OMP_PROC_BIND=close); bit-identical to the 1-thread run by construction (tests/test_oracle_cpu.py).  Half and a quarter of that are
    # timed too, the best one named.
    usable, one_socket, sockets = host_topology()
    sweep, k0, per, warm = {}, ns, 24, 8
    # (a box may grant this process far fewer CPUs than it has: the driver's MI355X box reports 256 and allows 2 -- the thread counts follow
    # what is usable: the physical cores of one socket, fractions of it, and every usable logical CPU when that is all there is)
    for T in sorted({one_socket, max(one_socket // 2, 1), max(one_socket // 4, 1), min(16, usable), min(usable, 2 * one_socket)}, reverse=True):
        o.set_threads(T)
        chunk = all_scans[k0:k0 + warm + per]
        if len(chunk) <= warm: break
        k0 += len(chunk)
        for t, ob in chunk[:warm]:                     # (the team's threads take hold of their columns)
            o.handle_observation(t, ob)
        t0 = time.perf_counter()
        for t, ob in chunk[warm:]:
            o.handle_observation(t, ob)
        sweep[T] = (len(chunk) - warm) / (time.perf_counter() - t0)
    best_T = max(sweep, key=sweep.get)
    t_all = 1.0 / sweep[best_T]
    n_all = k0 - ns
    o.set_threads(1)
    g2 = ReflectorEKFSLAM(S.options_for(sess), device=device)
    g2.set_state(st.time, st.mu, st.sigma, vt)
    err2 = []
    for (t, ob), po in zip(all_scans[:ns], poses):
        g2.handle_observation(t, ob)
        _, pg, _ = g2.pose()
        err2.append(float(np.sum((pg[:2] - po[:2]) ** 2)))
    g2.close()
    rmse = float(np.sqrt(np.mean(err2)))
    n = st.mu.shape[0]
    m = 2 * all_scans[0][1].shape[0]
    res = {"value": None, "unit": "updates/s", "cores": 1, "kind": "port",
           "structured_value": 1.0 / t_struct, "structured_sample": f"{ns} updates, O(n^2 m) algorithm, 1 thread",
           "structured_all_cores_value": 1.0 / t_all, "structured_all_cores": best_T,
           "structured_all_cores_sample": f"{per} further updates (behind {warm} untimed ones) per thread count, the same O(n^2 m) algorithm as one persistent "
                                          f"OpenMP team per update (pinned); the best of the thread counts tried is reported: {best_T} threads.  This process "
                                          f"may use {usable} logical CPU(s) = {one_socket} physical core(s) of one socket ({sockets} socket(s) visible; the "
                                          f"machine has {os.cpu_count()} logical CPUs: affinity mask and cgroup quota decide)",
           "usable_logical_cpus": usable, "physical_cores_one_socket": one_socket,
           "structured_best_value": sweep[best_T], "structured_best_threads": best_T,
           "structured_thread_sweep": {str(T): round(v, 2) for T, v in sorted(sweep.items())},
           "host_cores": os.cpu_count()}
    lit_scans = all_scans[ns + n_all:ns + n_all + nl]             # the scans that follow, times still increasing
    if len(lit_scans) == 0:
        res["sample"] = "no literal-mode update was timed (--cpu-literal-steps 0): value is null"
        return res, rmse
    o.set_mode(True)
    t0 = time.perf_counter()
    for t, ob in lit_scans:
        o.handle_observation(t, ob)
    t_lit = (time.perf_counter() - t0) / len(lit_scans)
    lit_flop = 6.0 * n ** 3 + 10.0 * m * n * n + 8.0 * m * m * n + 4.0 * m ** 3
    res.update({"value": 1.0 / t_lit,
                "sample": f"{len(lit_scans)} steady-state updates, oracle literal mode (dense O(n^3) sequence of the "
                          f"reference's Eigen expressions), from the GPU state at n={n}; 1 of {os.cpu_count()} host cores",
                "literal_s_per_update": t_lit, "literal_gflops": lit_flop / t_lit / 1e9})
    return res, rmse


def main():
    args = parse_args()
    respawn_under_torchrun(args)
    import torch

    from reflector_ekf_slam_amd import dist as D
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if os.environ.get("REKF_BENCH_FORCE_DIST") == "1" and "WORLD_SIZE" not in os.environ:
        # exercises the RCCL path at world size 1
        os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        import torch.distributed as dist_mod
        dist_mod.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        rank, world = 0, 1
    else:
        dist_mod, rank, local_rank, world = D.init("nccl")
    if world != args.gpus and "WORLD_SIZE" in os.environ and world > 1:
        args.gpus = world
    out = run_rank(args, dist_mod, rank, local_rank, world, device_sync=torch.cuda.synchronize)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_mod is not None:
        dist_mod.barrier()
        dist_mod.destroy_process_group()


if __name__ == "__main__":
    main()
