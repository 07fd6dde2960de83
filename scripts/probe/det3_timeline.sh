# usage (GPU box, repo root): bash scripts/probe/det3_timeline.sh  -- the 3D detector's call: per-call latency, kernel timeline (rocprofv3),
# host-side marks and in-kernel marks of a -DRDET_DEBUG_MARKS build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python scripts/gpu_bench_detectors.py 200 2>&1 | tail -2 | cut -c1-300
rm -rf gpurun_out/prof_det3
rocprofv3 --kernel-trace -d gpurun_out/prof_det3 -o det -- python scripts/gpu_dbg_det3d.py > /dev/null 2>&1
python scripts/rocpd_timeline.py gpurun_out/prof_det3/det_results.db 25 11
make -C reflector_ekf_slam_amd/csrc -B ../librdet.so HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -DRDET_DEBUG_MARKS" > /dev/null 2>&1
RDET3_HOST_MARKS=1 python scripts/gpu_dbg_det3d.py 16 link 2>&1 | grep -v "host us" | tail -${DET3_TL_ROWS:-40}
RDET3_HOST_MARKS=1 python scripts/gpu_dbg_det3d.py 2>&1 | grep "host us" | tail -3
