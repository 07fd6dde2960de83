#!/bin/bash
# usage (GPU box, repo root): bash scripts/gpu_ab_det3d.sh name1 "<-D flags 1>" name2 "<-D flags 2>" ...      ("" = the tree as it is)
# builds every variant of librdet.so into /tmp on the box and times them side by side in ONE process (scripts/gpu_ab_det3d.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared"
ARGS=""
while [ $# -gt 1 ]; do
  N=$1; F=$2; shift 2
  S=reflector_ekf_slam_amd/csrc
  if [ -f "$F" ]; then cp $F /tmp/librdet_$N.so; else hipcc $BASE $F $S/det2d.hip $S/det3d.hip -o /tmp/librdet_$N.so || { echo "build of $N failed"; continue; }; fi
  ARGS="$ARGS $N=/tmp/librdet_$N.so"
done
python scripts/gpu_ab_det3d.py --rings 16 $ARGS | tee -a gpurun_out/d3_ab.txt
python scripts/gpu_ab_det3d.py --rings 32 --rounds 6 --block 40 $ARGS | tee -a gpurun_out/d3_ab.txt
