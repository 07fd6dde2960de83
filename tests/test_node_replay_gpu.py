"""BASELINE.json configs[0] as a harness (-m gpu): a dump (rekf-dump-v1) replayed through the HIP path exactly as the reference's
Node drives its components -- LaserScan -> rdet2d -> rekf (capacity grows on demand, options.map_path on the second pass) ->
rgrid AddRangeData -> SaveReflectorResult -- against the same harness over the CPU oracle's components."""
import numpy as np
import pytest

from reflector_ekf_slam_amd import node_replay as NR
from reflector_ekf_slam_amd import synth
from reflector_ekf_slam_amd.ekf_slam import load_map_txt

pytestmark = pytest.mark.gpu


def test_dump_replay_matches_the_oracle_composition_and_the_saved_map_reloads(tmp_path, oracle_lib):
    from tests.oracle_node import oracle_backend
    cfg = synth.SessionConfig("dump_gpu", 40, 12, synth.DIFF, seed=31, speed=1.0, row_spacing=6.0)
    d = NR.synth_dump(str(tmp_path / "bag.npz"), cfg, max_scans=50, n_beams=2880)
    g = NR.replay(d, NR.hip_backend(max_landmarks=8))        # starts far too small: the capacity must grow on the way
    o = NR.replay(d, oracle_backend())
    assert len(g.log.observations) == len(o.log.observations) == d.scan_t.shape[0] - 1
    for (tg, cg), (to, co) in zip(g.log.observations, o.log.observations):
        assert tg == to and cg.shape == co.shape
        assert np.array_equal(cg, co)                        # detector centres: bit for bit (csrc/glibc_sincosf.h)
    assert g.slam.n == o.slam.n and g.slam.n > 3 + 2 * 10 and g.slam.max_landmarks >= (g.slam.n - 3) // 2 > 8
    assert g.slam.flags() == 0                               # no reflector was dropped
    sg, so = g.slam.GetState(), o.slam.GetState()
    assert np.abs(sg.mu - so.mu).max() < 1e-9                # identical observations in: FP64 round-off
    pg, po = np.array(g.log.path), np.array(o.log.path)
    assert pg.shape == po.shape and np.abs(pg - po).max() < 1e-9
    assert len(g.log.match_poses) == len(o.log.match_poses)
    both = [(a, b) for a, b in zip(g.log.match_poses, o.log.match_poses) if a is not None and b is not None]
    assert len(both) >= len(g.log.match_poses) - 2
    assert np.median([np.abs(a - b).max() for a, b in both]) < 1e-3        # scan-matcher poses: same candidates, LM refinement
    # SaveReflectorResult -> LoadMapFromTxtFile -> a second pass localises against the saved map (options.map_path on the GPU)
    path = g.SaveReflectorResult(str(tmp_path / "reflector_map"))
    m = load_map_txt(path)
    L = (g.slam.n - 3) // 2
    assert m.reflector_map_.shape == (L, 2)
    d2 = NR.read_dump(str(tmp_path / "bag.npz"))
    d2.meta["map_path"] = path
    g2 = NR.replay(d2, NR.hip_backend(max_landmarks=8))
    o2 = NR.replay(d2, oracle_backend())
    assert g2.slam.n == o2.slam.n <= g.slam.n                # most reflectors now match the MAP (covariance gate, Q3) and never enter the state
    mm = g2.slam.last_match()
    assert mm.map_obs_match_ids.shape[0] > 0
    assert np.abs(g2.slam.GetState().mu - o2.slam.GetState().mu).max() < 1e-9
