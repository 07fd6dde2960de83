"""Move the summaries a GPU profiling pass left under gpurun_out/profiles_<tag>/ into profiles/ and (re)write profiles/MANIFEST.json:
per summary the commit it was measured on and the sources whose kernels it describes (tests/test_profiles_cpu.py: stale as soon as
any later commit touches one of them).  Usage (this container, repo root, with the measured tree committed):
    python scripts/profiles_commit.py <tag> <commit> [--detectors]"""
import json, os, shutil, sys
tag, commit = sys.argv[1], sys.argv[2]
src = f"gpurun_out/profiles_{tag}"
EKF = ["reflector_ekf_slam_amd/csrc/ekf_kernels.hip", "reflector_ekf_slam_amd/csrc/rekf_api.hip", "reflector_ekf_slam_amd/csrc/ekf_dev.h"]
DET = ["reflector_ekf_slam_amd/csrc/det2d.hip", "reflector_ekf_slam_amd/csrc/det3d.hip", "reflector_ekf_slam_amd/csrc/glibc_sincosf.h",
       "reflector_ekf_slam_amd/csrc/host_visible.h"]
man_path = "profiles/MANIFEST.json"
man = json.load(open(man_path)) if os.path.exists(man_path) else {"summaries": {}}
man["note"] = ("per summary: the commit it was measured on and the sources whose kernels it describes; tests/test_profiles_cpu.py fails when a "
               "later commit touches one of them")
# entries of an older pass over the same sources are superseded
for f in sorted(os.listdir(src)):
    det = "detector" in f
    shutil.copy(os.path.join(src, f), os.path.join("profiles", f))
    srcs = DET if det else EKF
    for old in [k for k, e in man["summaries"].items() if e["sources"] == srcs and e["commit"] != commit and (k.split("_", 1)[-1] == f.split("_", 1)[-1])]:
        del man["summaries"][old]
    man["summaries"][f] = {"commit": commit, "sources": srcs}
# bench lines describe the whole tree: not guarded (they carry their own commit in the file name's tag)
for k in [k for k in man["summaries"] if "bench_" in k]:
    del man["summaries"][k]
json.dump(man, open(man_path, "w"), indent=1, sort_keys=True)
print(json.dumps(man, indent=1)[:2000])
