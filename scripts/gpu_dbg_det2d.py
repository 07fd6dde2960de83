"""In-kernel timeline of k_det2d (workgroup 0 = runs, workgroup 1 = first beam group) from a -DRDET_DEBUG_MARKS build:
  make -C reflector_ekf_slam_amd/csrc -B ../librdet.so HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -DRDET_DEBUG_MARKS"
GPU box: python scripts/gpu_dbg_det2d.py [beams]   (marks: shader-clock counter; host stamps in ns)"""
import ctypes as C, sys
sys.path.insert(0, ".")
from types import SimpleNamespace as NS
import numpy as np
from reflector_ekf_slam_amd import OdometryData, synth
from reflector_ekf_slam_amd.detect import LaserReflectorDetect, ReflectorDetectOptions

beams = int(sys.argv[1]) if len(sys.argv) > 1 else 3600
rng = np.random.Generator(np.random.PCG64(7))
lms = synth.make_world(synth.C2, rng)
pose = (float(lms[:, 0].mean()), float(lms[:, 1].mean()), 0.6)
scan = NS(**synth.make_laser_scan(lms, pose, 10.0, rng, n_beams=beams))
g = LaserReflectorDetect(ReflectorDetectOptions(), sensor_to_base_link=(0.13686, 0.0, 0.0))
for k in range(30):
    t = 9.5 + 0.02 * k
    g.HandleOdometryData(OdometryData(time=t, position=(0.5 * t, 0.0, 0.0), orientation=(1.0, 0.0, 0.0, 0.0),
                                      linear_velocity=(0.5, 0.0, 0.0), angular_velocity=(0.0, 0.0, 0.1)))
g._L.rdet2d_debug_marks.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
for rep in range(6):
    for _ in range(5):
        g.HandleLaserScan(scan)
    m = (C.c_ulonglong * 32)(); g._L.rdet2d_debug_marks(g._h, m)
    m = list(m); t0 = min(m[0], m[16])
    ghz = 2.4                                                     # shader clock assumed for the conversion
    print("runs  wg:", [round((x - m[0]) / ghz * 1e-3, 2) for x in m[0:12]], "us")
    print("beams wg:", [round((x - m[16]) / ghz * 1e-3, 2) for x in m[16:23]], "us (own clock)")
    print("host: scan written %.2f, launched %.2f, head seen %.2f, centres out %.2f us" % tuple(x * 1e-3 for x in m[24:28]))
