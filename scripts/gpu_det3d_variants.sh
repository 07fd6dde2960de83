#!/bin/bash
# usage (GPU box, repo root): bash scripts/gpu_det3d_variants.sh "<-D flags of variant 1>" "<-D flags of variant 2>" ...
# A/B of compile-time variants of the 3D detector on one box: each is built into the box's scratch copy, checked against the oracle and timed
# (16- and 32-ring clouds).  "" = the tree as it is.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950"
for V in "$@"; do
  make -C reflector_ekf_slam_amd/csrc -B ../librdet.so HIPFLAGS="$BASE $V" > /dev/null 2>&1 || { echo "variant [$V]: build failed"; continue; }
  echo "== variant [$V]" | tee -a gpurun_out/d3_variants.txt
  for R in 1 2; do
    python scripts/gpu_det3d_only.py 300 16 | tee -a gpurun_out/d3_variants.txt
  done
  python scripts/gpu_det3d_only.py 150 32 | tee -a gpurun_out/d3_variants.txt
done
