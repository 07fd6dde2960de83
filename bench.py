#!/usr/bin/env python
"""bench.py -- EKF updates/s at N=1024 landmarks x 32 observations/scan (BASELINE.json
configs[2], "C3") on MI355X, one filter session per GPU.

A "step" is one steady-state HandleObservationMessage (predict + ReflectorMatch +
EKF update; reference reflector_ekf_slam.cc:229-309) on a filter whose covariance
(n = 2051, 33.6 MB FP64) is resident in HBM.  The map-building warm-up (the augment
path) runs untimed before it.  Inputs per step are the scan's 32 float32 points,
passed by value with the launch; nothing else crosses PCIe in the timed region.

    python bench.py --gpus N --steps K --warmup W

For N > 1 the driver launches one rank per GPU with torch.distributed.run; ranks
run independent sessions (seed + rank, BASELINE.json configs[4]) -- "replicas
only", no data-path collective -- and rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      the P -= K (H P) kernel (k_downdate): algorithmic bytes per launch
                (16 n^2 + 8 n (3+m)) / its average launch time,
                measured with hipEvents on the handle's stream in an instrumented
                pass over the same K steps.
  cpu_baseline  the CPU oracle (oracle/ekf_oracle.c, "port") in LITERAL mode -- the
                dense operation sequence the reference's Eigen expressions execute --
                timed on this box's host, 1 thread, on a bounded sample of the same
                steady-state steps, started from the GPU's own state.
  multi_session aggregate updates/s of 4 independent sessions sharing GPU 0 (a secondary figure for
                fleet serving; `value` stays the single-session rate).
"""
from __future__ import annotations

import os

# The multi_session leg runs 4 sessions next to the main handle: five HIP streams.  ROCm maps streams onto 4 hardware
# queues per process by default (two sessions would share one and serialise); must be set before the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_MFMA_PEAK_TF = 78.6       # MI355X datasheet FP64 matrix (v_mfma_f64_16x16x4_f64: 32 FLOP/clk/SIMD)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", default="C3", choices=["C2", "C3", "C4"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--multi-sessions", type=int, default=4,
                    help="also report the aggregate rate of this many independent sessions sharing GPU 0 (0: skip)")
    ap.add_argument("--cpu-literal-steps", type=int, default=3)
    ap.add_argument("--cpu-structured-steps", type=int, default=60)
    return ap.parse_args()


def main():
    args = parse_args()
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("REKF_BENCH_FORCE_DIST") == "1":   # the env var exercises the RCCL path at world size 1
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    from reflector_ekf_slam_amd import ReflectorEKFSLAM, synth
    from reflector_ekf_slam_amd import session as S
    import dataclasses

    base = getattr(synth, args.config)
    cfg = dataclasses.replace(base, seed=base.seed + (10 + rank if world > 1 else 0),
                              name=base.name + (f"_rank{rank}" if world > 1 else ""))
    sess = synth.make_session(cfg)
    L = cfg.n_landmarks
    n_expect = 3 + 2 * L
    ekf = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=L, device=local_rank)

    # ---- untimed: build the map (augment path) ---------------------------------
    t0 = time.time()
    S.replay(sess, ekf)
    ekf.sync()
    n = ekf.n
    map_build_s = time.time() - t0
    if n != n_expect:
        raise SystemExit(f"warm-up ended with n={n}, expected {n_expect}")

    steady = synth.steady_state_scans(sess, args.warmup + 2 * args.steps + 8)
    m = 2 * steady[0][1].shape[0]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ekf.sync()

    it = iter(steady)
    for _ in range(args.warmup):
        t, ob = next(it)
        ekf.handle_observation(t, ob)

    # ---- timed region: exactly K steps --------------------------------------------
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        t, ob = next(it)
        ekf.handle_observation(t, ob)
    ekf.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    if dist is not None:
        dist.barrier()
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed_max = float(tt.item())
    else:
        elapsed_max = elapsed
    mm = ekf.last_match()
    assert len(mm.new_ids) == 0 and len(mm.state_obs_match_ids) == m // 2, "not steady state"
    assert ekf.n == n_expect

    # ---- secondary figure: several sessions on this GPU (before the instrumented pass litters the runtime with events)
    ms_result = multi_session(args, cfg, sess, local_rank) if (rank == 0 and args.multi_sessions > 1) else None

    # ---- instrumented pass: per-kernel hipEvent timing over the same number of steps
    state_for_cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        state_for_cpu = ekf.GetState()
    ekf.profile_reset()
    ekf.profile(True)
    cpu_scans = []
    for k in range(args.steps):
        t, ob = next(it)
        ekf.handle_observation(t, ob)
        if k < args.cpu_literal_steps + args.cpu_structured_steps:
            cpu_scans.append((t, ob))
    ekf.profile(False)
    prof = ekf.profile_read()
    kernel_us = {k: (v[0] / v[1] if v[1] else None) for k, v in prof.items()}
    # The roofline kernel once more, without per-launch brackets: `steps` back-to-back launches between ONE
    # pair of hipEvents on the handle's stream (operands = what the last scan left in HBM).  A bracket
    # around every launch adds 2-3 us of command-processor time to each reading (an empty bracket reads
    # 4-5 us); this figure does not, and agrees with rocprofv3 --kernel-trace (profiles/).  The filter
    # state is meaningless afterwards; nothing below uses this handle again.
    dd_batched_us = ekf.time_kernel("downdate", reps=max(args.steps, 100))

    out = None
    if rank == 0:
        # roofline.achieved uses the un-bracketed event measurement (dd_batched_us, above); the per-launch
        # bracket reading, the empty-bracket reading and the rocprof average of the last committed
        # profile are reported beside it.
        ev_overhead = kernel_us.get("empty") or 0.0
        dd_bracket_us = kernel_us["downdate"]
        dd_us = dd_batched_us
        rocprof_us = None
        try:
            rocprof_us = json.load(open(os.path.join(ROOT, "profiles", "kernel_avg_us.json"))).get("k_downdate")
        except Exception:
            pass
        bytes_alg = 16.0 * n * n + 8.0 * n * (3 + m)          # SURVEY.md 8(d) BYTES_alg(n, m)
        flop_k7 = 2.0 * n * n * m
        achieved = bytes_alg / (dd_us * 1e-6) / 1e9
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_downdate.json")
        if os.path.exists(pmc_path):
            try:
                traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "EKF updates/s at N=1024 landmarks, 32 obs/scan; pose RMSE vs reference",
            "value": world * args.steps / elapsed_max,
            "unit": "updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{base.name}: synthetic 2D session, L={L} landmarks (n={n}), "
                                   f"{m // 2} matched observations/scan (m={m}), diff-drive odometry, "
                                   "steady-state HandleObservationMessage",
                       "sessions": world, "parallelism": "replicas: one independent session per GPU",
                       "map_build_s": round(map_build_s, 3)},
            "roofline": {"kernel": "k_downdate (P -= K (H P), FP64 MFMA 16x16x4)", "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "bytes_per_launch": bytes_alg, "avg_launch_us": dd_us,
                         "avg_launch_us_method": "back-to-back launches between one hipEvent pair on the handle's stream",
                         "per_launch_bracket_us": dd_bracket_us, "empty_event_bracket_us": ev_overhead,
                         "rocprof_avg_launch_us": rocprof_us,
                         "mfma": {"achieved_tflops": flop_k7 / (dd_us * 1e-6) / 1e12,
                                  "peak_tflops": FP64_MFMA_PEAK_TF,
                                  "frac": flop_k7 / (dd_us * 1e-6) / 1e12 / FP64_MFMA_PEAK_TF}},
            "kernel_us": {k: (round(v, 3) if v is not None else None) for k, v in kernel_us.items()},
        }
        if ms_result is not None:
            out["multi_session"] = ms_result
        if not args.no_cpu_baseline:
            out["cpu_baseline"], out["pose_rmse_vs_oracle_m"] = cpu_baseline(args, cfg, sess, state_for_cpu,
                                                                              cpu_scans, ekf)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def multi_session(args, cfg, sess, device):
    """Secondary figure (never `value`): S independent filter sessions of the same workload sharing ONE GPU, one handle
    = one HIP stream each, fed round-robin by this host thread.  A single session is a latency-bound chain of five
    kernels (one of them a single workgroup), so sessions interleave on the device: the aggregate rate is what a fleet
    server gets per GPU.  Every session runs the same scans and must end bit-identical."""
    from reflector_ekf_slam_amd import ReflectorEKFSLAM, synth
    from reflector_ekf_slam_amd import session as S
    ns = args.multi_sessions
    steps = min(args.steps, 1000)
    scans = synth.steady_state_scans(sess, 100 + steps)
    handles = []
    for _ in range(ns):
        g = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, device=device)
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


def cpu_baseline(args, cfg, sess, st, scans, ekf):
    """Time the CPU oracle on a bounded sample of the same steady-state steps.

    The oracle is started from the state the GPU had at the beginning of the
    instrumented pass (st) and fed the same scans; its poses are also compared with
    the GPU's (pose RMSE over the structured-mode sample)."""
    from oracle.binding import OracleEKF
    o = OracleEKF(cfg.odom_model, sess.init_time, sess.init_pose, cfg.sigma_v ** 2, cfg.sigma_w ** 2,
                  cfg.sigma_obs ** 2)
    vt = sess.odom[np.nonzero(sess.ev_type == 0)[0][-1]]
    o.set_state(st.time, st.mu, st.sigma, vt)
    # structured mode first (also the parity sample), then literal on the following scans
    ns = min(args.cpu_structured_steps, len(scans))
    t0 = time.perf_counter()
    poses = []
    for t, ob in scans[:ns]:
        o.handle_observation(t, ob)
        poses.append(o.mu()[:3].copy())
    t_struct = (time.perf_counter() - t0) / max(ns, 1)
    # GPU poses for the same steps: replay on a scratch handle from the same state
    from reflector_ekf_slam_amd import ReflectorEKFSLAM
    from reflector_ekf_slam_amd import session as S
    g2 = ReflectorEKFSLAM(S.options_for(sess), max_landmarks=cfg.n_landmarks, device=int(os.environ.get("LOCAL_RANK", "0")))
    g2.set_state(st.time, st.mu, st.sigma, vt)
    err2 = []
    for (t, ob), po in zip(scans[:ns], poses):
        g2.handle_observation(t, ob)
        _, pg, _ = g2.pose()
        err2.append(float(np.sum((pg[:2] - po[:2]) ** 2)))
    g2.close()
    rmse = float(np.sqrt(np.mean(err2))) if err2 else None
    o.set_mode(True)
    lit_scans = scans[ns:ns + args.cpu_literal_steps]      # the scans that follow, times still increasing
    t0 = time.perf_counter()
    for t, ob in lit_scans:
        o.handle_observation(t, ob)
    t_lit = (time.perf_counter() - t0) / max(len(lit_scans), 1)
    n = st.mu.shape[0]
    m = 2 * scans[0][1].shape[0]
    lit_flop = 6.0 * n ** 3 + 10.0 * m * n * n + 8.0 * m * m * n + 4.0 * m ** 3
    return ({"value": 1.0 / t_lit, "unit": "updates/s", "cores": 1, "kind": "port",
             "sample": f"{len(lit_scans)} steady-state updates, oracle literal mode (dense O(n^3) sequence of the "
                       f"reference's Eigen expressions), from the GPU state at n={n}; host has {os.cpu_count()} cores",
             "literal_s_per_update": t_lit, "literal_gflops": lit_flop / t_lit / 1e9,
             "structured_value": 1.0 / t_struct, "structured_sample": f"{ns} updates, O(n^2 m) algorithm, 1 thread",
             "host_cores": os.cpu_count()}, rmse)


if __name__ == "__main__":
    main()
