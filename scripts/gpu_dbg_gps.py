import sys
sys.path.insert(0, ".")
import numpy as np
from tests.helpers import make_gpu
K = int(sys.argv[1]); gps = sys.argv[2] == "1"
rng = np.random.default_rng(1)
g = make_gpu(0, 0.0, np.zeros(3), 0.0025, 0.0064, 0.0025, 64)
pts = rng.uniform(-8, 8, size=(K, 2)).astype(np.float32)
pts = pts[np.argsort(pts[:, 0])]
g.handle_observation(0.0, pts)
g.handle_odometry(0.1, 0.5, 0.0, 0.1)
g.handle_observation(0.2, pts + 0.01, (0.05, 0.0, 0.01) if gps else None)
g.sync(); m = g.last_match()
print("K", K, "gps", gps, "OK n", g.n, "state", len(m.state_obs_match_ids), flush=True)
