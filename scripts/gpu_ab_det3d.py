"""In-process A/B of builds of librdet.so (3D detector): every library is loaded side by side, each gets its own handle, and the timed blocks
alternate between them, so box-to-box and process-to-process differences (1-2 us on this call) cancel.
Usage (GPU box): python scripts/gpu_ab_det3d.py [--rings 16] [--rounds 8] [--block 60] name=path.so [name=path.so ...]"""
import ctypes as C, json, sys, time
sys.path.insert(0, ".")
import numpy as np
from reflector_ekf_slam_amd import synth
from reflector_ekf_slam_amd.detect import Rdet3dOptions, MAX_CENTERS
from oracle.binding import oracle_detect3d

args = sys.argv[1:]
rings, rounds, block = 16, 8, 60
libs = []
while args:
    a = args.pop(0)
    if a == "--rings": rings = int(args.pop(0))
    elif a == "--rounds": rounds = int(args.pop(0))
    elif a == "--block": block = int(args.pop(0))
    else: libs.append(a.split("=", 1))
rng = np.random.Generator(np.random.PCG64(7))
for i in range(3):
    synth.make_world(synth.C2 if i < 2 else synth.C3, rng)
lms = synth.make_world(synth.C4, rng)
pose = (float(lms[:, 0].mean()), float(lms[:, 1].mean()), 0.3)
cloud = np.ascontiguousarray(synth.make_point_cloud(lms, pose, rng, rings=rings, n_az=1800), np.float32)
co, m1, m2 = oracle_detect3d(cloud)
vp = C.c_void_p


class Det:
    def __init__(self, path):
        L = self.L = C.CDLL(path)
        L.rdet3d_create.argtypes = [C.POINTER(Rdet3dOptions), C.POINTER(C.c_double), C.c_int, C.c_int, C.POINTER(vp)]
        L.rdet3d_handle_cloud.argtypes = [vp, C.c_double, vp, C.c_int, vp, C.c_int, vp, vp]
        L.rdet3d_destroy.argtypes = [vp]
        L.rdet3d_destroy.restype = None
        self.h = vp()
        s2b = (C.c_double * 3)(0.0, 0.0, 0.0)
        o = Rdet3dOptions(150.0)
        from reflector_ekf_slam_amd.detect import PointCloudOptions
        o = Rdet3dOptions(PointCloudOptions().intensity_min)
        assert L.rdet3d_create(C.byref(o), s2b, 65536, 0, C.byref(self.h)) == 0
        self.out = np.zeros((MAX_CENTERS, 2), np.float32)
        self.K = C.c_int(0)
        self.t = C.c_double(0)

    def call(self):
        rc = self.L.rdet3d_handle_cloud(self.h, 1.0, cloud.ctypes.data, cloud.shape[0], self.out.ctypes.data, MAX_CENTERS, C.byref(self.K), C.byref(self.t))
        assert rc == 0, rc
        return self.out[: self.K.value]


dets = [(n, Det(p)) for n, p in libs]
for n, d in dets:
    r = d.call().copy()
    ok = r.shape == co.shape and bool((r == co).all())
    print("%-14s identical_to_oracle %s (%d centres, gate %d, sor %d)" % (n, ok, r.shape[0], m1, m2))
    for _ in range(30):
        d.call()
ts = {n: [] for n, _ in dets}
for r in range(rounds):
    for n, d in dets[::1 if r % 2 == 0 else -1]:
        for _ in range(5):
            d.call()
        for _ in range(block):
            t0 = time.perf_counter()
            d.call()
            ts[n].append((time.perf_counter() - t0) * 1e6)
base = None
for n, _ in dets:
    v = np.array(ts[n])
    med = float(np.median(v))
    base = med if base is None else base
    print(json.dumps({"variant": n, "rings": rings, "call_us_median": round(med, 2), "p10": round(float(np.percentile(v, 10)), 2),
                      "p90": round(float(np.percentile(v, 90)), 2), "vs_first": round(med - base, 2)}))
