#!/bin/bash
# usage (GPU box, repo root): bash scripts/gpu_det3d_stage.sh <tag> [marks]
# the 3D detector alone: call latency (16- and 32-ring clouds), rocprofv3 kernel stats of the 16-ring cloud, and -- with "marks" -- the in-kernel
# timelines of a -DRDET_DEBUG_MARKS build (built on the box, in the box's scratch copy).
TAG=${1:-d3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python scripts/gpu_det3d_only.py 400 16 | tee gpurun_out/${TAG}_call.jsonl
python scripts/gpu_det3d_only.py 200 32 | tee -a gpurun_out/${TAG}_call.jsonl
rm -rf gpurun_out/prof_d3
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_d3 -o d3 -- python scripts/gpu_det3d_only.py 100 16 > /dev/null 2>&1
python scripts/rocpd_stats.py gpurun_out/prof_d3/d3_results.db > gpurun_out/${TAG}_kernel_stats.txt 2>&1
cat gpurun_out/${TAG}_kernel_stats.txt | cut -c1-60,60-130 | head -16
if [ "$2" = "marks" ]; then
  make -C reflector_ekf_slam_amd/csrc -B ../librdet.so HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -DRDET_DEBUG_MARKS" > /dev/null 2>&1
  RDET3_HOST_MARKS=1 python scripts/gpu_dbg_det3d.py 16 link 2>&1 | tail -80 | tee gpurun_out/${TAG}_marks.txt
fi
