"""CPU side of the config-1 harness (reflector_ekf_slam_amd/node_replay.py): the dump schema round-trips, the replay order is
the node's, and the whole message flow runs over the CPU oracle's components (no GPU): scan -> 2D detect -> EKF ->
AddRangeData -> SaveReflectorResult -> the saved txt map loads again."""
import numpy as np
import pytest

from reflector_ekf_slam_amd import node_replay as NR
from reflector_ekf_slam_amd import synth
from reflector_ekf_slam_amd.ekf_slam import load_map_txt


def _cfg():
    return synth.SessionConfig("dump_t", 24, 8, synth.DIFF, seed=31, speed=1.0, row_spacing=6.0)


def test_dump_round_trip_and_event_order(tmp_path):
    p = str(tmp_path / "d.npz")
    d = NR.synth_dump(p, _cfg(), max_scans=6, n_beams=720)
    d2 = NR.read_dump(p)
    assert d2.meta["schema"] == NR.SCHEMA and d2.meta["odom_model"] == "diff"
    for a in ("odom_t", "odom_pose", "odom_twist", "scan_t", "scan_params", "scan_off", "ranges", "intensities"):
        assert np.array_equal(getattr(d, a), getattr(d2, a))
    assert d.scan_t.shape[0] == 6 and d.scan(2).ranges.shape[0] == 720 and d.odom_pose.shape[1] == 4
    ev = d.events()
    stamps = [d.odom_t[i] if k == "odom" else d.scan_t[i] for k, i in ev]
    assert stamps == sorted(stamps) and len(ev) == d.odom_t.shape[0] + d.scan_t.shape[0]
    np.savez(str(tmp_path / "bad.npz"), meta=np.array('{"schema": "something-else"}'))
    with pytest.raises(ValueError):
        NR.read_dump(str(tmp_path / "bad.npz"))


def test_node_flow_over_the_oracle_components(tmp_path, oracle_lib):
    from tests.oracle_node import oracle_backend
    d = NR.synth_dump(str(tmp_path / "d.npz"), _cfg(), max_scans=14, n_beams=1440)
    node = NR.replay(d, oracle_backend())
    assert len(node.log.observations) == d.scan_t.shape[0] - 1                  # the first scan only constructs the EKF (Q11)
    assert node.slam.n > 3 + 2 * 3                                              # reflectors were detected and mapped
    assert sum(p is not None for p in node.log.match_poses) >= len(node.log.match_poses) - 1
    assert len(node.log.path) >= d.odom_t.shape[0] // 2
    path = node.SaveReflectorResult(str(tmp_path / "reflector_map"))
    m = load_map_txt(path)                                                      # the reference's bytes (leading comma and all) load
    L = (node.slam.n - 3) // 2
    assert m.reflector_map_.shape == (L, 2) and m.reflector_map_coviarance_.shape == (L, 2, 2)
    st = node.slam.GetState()
    assert np.allclose(m.reflector_map_, st.mu[3:].reshape(-1, 2), rtol=1e-5, atol=1e-5)    # 6 significant digits, like the reference
