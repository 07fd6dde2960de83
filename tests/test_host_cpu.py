"""CPU suite: the C-ABI library loads and exports every declared symbol, the synthetic
generator is deterministic, the txt map format round-trips, and the N>1 harness path
works over gloo with world_size 2.  No compute calls into the HIP library (no GPU here)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:rekf|rdet2d|rdet3d|rdet|rgrid)_[a-z0-9_]+)\s*\(", text)))


@pytest.mark.parametrize("header,lib", [("rekf.h", "librekf.so"), ("rekf_debug.h", "librekf.so"), ("rdet.h", "librdet.so"), ("rgrid.h", "librgrid.so")])
def test_abi_library_exports_every_declared_symbol(header, lib):
    if not os.path.exists(os.path.join(ROOT, "include", header)):
        pytest.skip(f"{header} not part of this build yet")
    import __graft_entry__
    path = os.path.join(ROOT, "reflector_ekf_slam_amd", lib)
    if not os.path.exists(path):
        __graft_entry__.build()
    L = ctypes.CDLL(path)
    syms = _declared_symbols(header)
    assert len(syms) >= (6 if header == "rekf_debug.h" else 9)
    for s in syms:
        assert hasattr(L, s), f"{lib} does not export {s} declared in include/{header}"


def test_abi_version_and_strerror():
    from reflector_ekf_slam_amd import _lib
    L = _lib.rekf()
    text = open(os.path.join(ROOT, "include", "rekf.h")).read()
    macro = int(re.search(r"#define\s+REKF_ABI_VERSION\s+(\d+)", text).group(1))
    assert L.rekf_abi_version() == macro == _lib.REKF_ABI_VERSION      # header, built library and Python loader agree
    assert L.rekf_strerror(0) == b"ok"
    assert b"observations" in L.rekf_strerror(-3)
    # null handle -> error code, never a crash
    assert L.rekf_sync(None) == -1
    assert L.rekf_handle_odometry(None, 0.0, 0.0, 0.0, 0.0) == -1


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under reflector_ekf_slam_amd/ may reference it."""
    pkg = os.path.join(ROOT, "reflector_ekf_slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                for pat in (r"^\s*(from|import)\s+oracle", r"liboracle", r"\boekf_", r"\bogrid_", r"oracle/"):
                    assert not re.search(pat, txt, flags=re.M), f"{f} reaches into the oracle ({pat})"


def test_synth_is_deterministic_and_covers_the_world():
    from reflector_ekf_slam_amd import synth
    a = synth.make_session(synth.C2)
    b = synth.make_session(synth.C2)
    assert np.array_equal(a.obs, b.obs) and np.array_equal(a.odom, b.odom)
    assert a.obs.dtype == np.float32
    assert len(set(a.obs_truth_id.tolist())) == synth.C2.n_landmarks        # every reflector gets seen
    # nearest-neighbour distance stays far above twice the 0.6 m association gate
    d = np.linalg.norm(a.landmarks[:, None] - a.landmarks[None], axis=-1) + 1e9 * np.eye(a.landmarks.shape[0])
    assert d.min() >= 2.0
    st = synth.steady_state_scans(a, 3)
    assert st[0][1].shape == (synth.C2.obs_per_scan, 2) and st[1][0] > st[0][0]


def test_golden_inputs_match_the_generator(golden_dir):
    """The committed fixtures carry their own inputs; they must equal what synth generates today."""
    from reflector_ekf_slam_amd import synth
    from tests.golden.make_golden import CASES
    g = np.load(os.path.join(golden_dir, "diff_L24_obs8.npz"))
    cfg, max_scans, _, _ = CASES["diff_L24_obs8"]
    s = synth.make_session(cfg, max_scans=max_scans)
    assert np.array_equal(g["obs"], s.obs) and np.array_equal(g["odom"], s.odom)


def test_map_txt_round_trip(tmp_path):
    from reflector_ekf_slam_amd.ekf_slam import Map, State, load_map_txt, save_map_txt
    mu = np.array([0.0, 0.0, 0.0, 1.5, 2.5, -3.25, 4.0])
    sig = np.eye(7) * 0.01
    sig[3, 4] = sig[4, 3] = 0.002
    p = tmp_path / "map.txt"
    save_map_txt(str(p), State(0.0, mu, sig))
    m = load_map_txt(str(p))
    assert m.reflector_map_.shape == (2, 2) and m.reflector_map_.dtype == np.float32
    assert np.allclose(m.reflector_map_, [[1.5, 2.5], [-3.25, 4.0]])
    assert np.allclose(m.reflector_map_coviarance_[0], [[0.01, 0.002], [0.002, 0.01]])
    assert load_map_txt(str(tmp_path / "missing.txt")).reflector_map_.shape[0] == 0     # cc:45-46
    (tmp_path / "bad.txt").write_text("1,2,3,4\n1,2,3\n")
    assert load_map_txt(str(tmp_path / "bad.txt")).reflector_map_.shape[0] == 0          # cc:74-79


def test_map_txt_reference_bytes(tmp_path):
    """reference_bytes=True: the very text std::ofstream << writes (src/ros_node.cc:86-136): %g / 6 significant
    digits, pre-loaded points from their float32 values, and the reference's "," in front of the new landmarks."""
    from reflector_ekf_slam_amd.ekf_slam import Map, State, load_map_txt, save_map_txt
    mu = np.array([0.0, 0.0, 0.0, 1.23456789, -2.5, 1e-7, 123456789.0])
    sig = np.zeros((7, 7))
    sig[3:5, 3:5] = [[0.0123456789, 1e-5], [2e-5, 0.5]]
    sig[5:7, 5:7] = [[1.0, 0.0], [0.0, 3.0]]
    p = tmp_path / "ref.txt"
    save_map_txt(str(p), State(0.0, mu, sig), reference_bytes=True)
    assert p.read_text() == ",1.23457,-2.5,1e-07,1.23457e+08\n,0.0123457,1e-05,2e-05,0.5,1,0,0,3\n"
    m = load_map_txt(str(p))                       # our loader skips the empty leading field
    assert m.reflector_map_.shape == (2, 2) and abs(m.reflector_map_[0, 0] - 1.23457) < 1e-6
    pre = Map(np.array([[0.1, 7.0]], np.float32), np.array([[[0.25, 0.0], [0.0, 0.125]]]))
    save_map_txt(str(p), State(0.0, mu[:5], sig[:5, :5]), loaded=pre, reference_bytes=True)
    assert p.read_text() == "0.1,7,1.23457,-2.5\n0.25,0,0,0.125,0.0123457,1e-05,2e-05,0.5\n"
    save_map_txt(str(p), State(0.0, mu[:3], sig[:3, :3]), loaded=pre, reference_bytes=True)
    assert p.read_text() == "0.1,7\n0.25,0,0,0.125\n"


_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
from reflector_ekf_slam_amd import dist as D, synth, session as S
from oracle.binding import OracleEKF
dist, rank, local_rank, world = D.init("gloo")
cfg = synth.SessionConfig("w", 12, 6, synth.DIFF, seed=100 + rank, speed=1.0, row_spacing=6.0)
sess = synth.make_session(cfg, max_scans=40)
f = OracleEKF(cfg.odom_model, sess.init_time, sess.init_pose, cfg.sigma_v**2, cfg.sigma_w**2, cfg.sigma_obs**2)
D.barrier(dist)
steps = S.replay(sess, f)
elapsed = 0.5 + 0.25 * rank
emax = D.max_over_ranks(dist, elapsed)
mu = f.mu()
recs = D.gather_records(dist, dict(steps=steps, elapsed_s=elapsed, final_n=f.n, pose_x=mu[0], pose_y=mu[1],
                                   pose_theta=mu[2], max_abs_err=0.0, seed=cfg.seed))
if rank == 0:
    assert len(recs) == world and [int(r["seed"]) for r in recs] == [100, 101]
    assert abs(emax - 0.75) < 1e-12
    assert recs[0]["pose_x"] != recs[1]["pose_x"]          # independent sessions
    agg = D.aggregate_updates_per_s(recs, emax)
    assert abs(agg - (recs[0]["steps"] + recs[1]["steps"]) / 0.75) < 1e-9
    print("GLOO_OK", agg)
D.barrier(dist)
dist.destroy_process_group()
"""


def test_session_per_rank_harness_over_gloo_world_size_2(tmp_path, oracle_lib):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "GLOO_OK" in out.stdout


_BENCH_WORKER = """
import os, sys, json
sys.path.insert(0, {root!r})
import numpy as np
import bench
from reflector_ekf_slam_amd import dist as D
from oracle.binding import OracleEKF

class Stub:
    # the CPU oracle behind the snake_case filter interface bench.run_rank drives (test only)
    def __init__(self, cfg, sess, device):
        self.o = OracleEKF(cfg.odom_model, sess.init_time, sess.init_pose, cfg.sigma_v**2, cfg.sigma_w**2, cfg.sigma_obs**2)
    def handle_odometry(self, *a): self.o.handle_odometry(*a)
    def handle_observation(self, *a): self.o.handle_observation(*a)
    def sync(self): pass
    def mu(self): return self.o.mu()
    def last_match(self): return self.o.last_match()
    def GetState(self):
        from types import SimpleNamespace
        mu, P = self.o.state()
        return SimpleNamespace(time=self.o.time, mu=mu, sigma=P)
    n = property(lambda self: self.o.n)

dist, rank, local_rank, world = D.init("gloo")
args = bench.parse_args(["--gpus", "2", "--config", "T0", "--steps", "7", "--warmup", "3"])
out = bench.run_rank(args, dist, rank, local_rank, world, make_filter=Stub, full=False)
if rank == 0:
    assert out["n_gpus"] == 2 and out["steps"] == 7 and out["warmup"] == 3 and out["scaling"] == "weak"
    assert out["metric"].startswith("EKF updates/s at N=12 landmarks, 6 obs/scan")
    r = out["ranks"]
    assert len(r["seeds"]) == 2 and r["seeds"][0] != r["seeds"][1] and r["final_n"] == [27, 27]
    assert r["updates_per_s_min"] <= r["updates_per_s_median"] <= r["updates_per_s_max"]
    # every rank's record carries its parity figure (SURVEY 8(e)); here the oracle replays against itself on both ranks
    assert r["max_abs_err_vs_oracle"] == [0.0, 0.0]
    # whole-job value = all ranks' steps / the slowest rank's time  <=  sum of the per-rank rates
    assert 0 < out["value"] <= 2 * r["updates_per_s_max"] * (1 + 1e-9)
    assert abs(out["value"] - 2 * 7 / (out["ms_per_step"] * 1e-3 * 7)) < 1e-6 * out["value"]
    print("BENCH_GLOO_OK", json.dumps(out)[:200])
else:
    assert out is None
D.barrier(dist)
dist.destroy_process_group()
"""


def test_bench_rank_function_over_gloo_world_size_2(tmp_path, oracle_lib):
    """bench.py's own per-rank function (timed region, MAX over ranks, record all-gather, aggregate + per-rank
    min/median) at world size 2 over gloo, with the CPU oracle standing in for the HIP handle."""
    script = tmp_path / "bench_worker.py"
    script.write_text(_BENCH_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29519", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "BENCH_GLOO_OK" in out.stdout


def test_bench_gpus_flag_respawns_under_torchrun(monkeypatch):
    """`python bench.py --gpus 4` outside torchrun must re-exec itself as 4 ranks; inside torchrun it must not."""
    import bench
    seen = {}
    monkeypatch.setattr(os, "execvpe", lambda f, a, e: seen.update(cmd=a, env=e))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    bench.respawn_under_torchrun(bench.parse_args(["--gpus", "4", "--steps", "20", "--warmup", "5"]))
    assert "--nproc-per-node=4" in seen["cmd"] and seen["cmd"][-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]
    assert "127.0.0.1" in seen["cmd"] and seen["env"]["MASTER_ADDR"] == "127.0.0.1"
    seen.clear()
    monkeypatch.setenv("WORLD_SIZE", "4")
    bench.respawn_under_torchrun(bench.parse_args(["--gpus", "4"]))
    assert not seen
    monkeypatch.delenv("WORLD_SIZE")
    bench.respawn_under_torchrun(bench.parse_args(["--gpus", "1"]))
    assert not seen


@pytest.mark.parametrize("header", ["rekf.h", "rekf_debug.h", "rdet.h", "rgrid.h"])
def test_c_headers_are_plain_c99(header, tmp_path):
    """The drop-in boundary is a C ABI: every public header must compile as pedantic C99 on its own."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text(f'#include "{header}"\nint main(void) {{ return 0; }}\n')
    out = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                          "-o", str(tmp_path / "t.o")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_rgrid_abi_version_is_consistent():
    """include/rgrid.h, the built library and the Python loader agree on the ABI version."""
    from reflector_ekf_slam_amd import grid
    text = open(os.path.join(ROOT, "include", "rgrid.h")).read()
    macro = int(re.search(r"#define\s+RGRID_ABI_VERSION\s+(\d+)", text).group(1))
    assert macro == grid.RGRID_ABI_VERSION == grid._lib_rgrid().rgrid_abi_version()
