# usage (GPU box, repo root): bash scripts/probe/gj_marks.sh -- where wave 0 of mid workgroup 1 spends the inverse: marks in front of and behind each of
# its four block-step barriers (-DREKF_DEBUG_TIMING -DREKF_DEBUG_GJ build into scripts/probe/librekf_dbg.so)
cd $GRAFT_REPO_ROOT/reflector_ekf_slam_amd/csrc
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -mllvm -amdgpu-kernarg-preload-count=8 -DREKF_DEBUG_TIMING -DREKF_DEBUG_GJ -shared ekf_kernels.hip rekf_api.hip -o ../../scripts/probe/librekf_dbg.so 2>&1 | grep -E "error" -A3
cd $GRAFT_REPO_ROOT
python scripts/probe/scan_marks.py scripts/probe/librekf_dbg.so C3 2>&1 | tail -9
