#!/bin/bash
# usage (GPU box, repo root): bash scripts/gpu_ab_rekf.sh [--tests] name1 "<-D flags 1>" name2 "<-D flags 2>" ...     ("" = the tree as it is)
# A/B of compile-time variants of librekf.so: each variant is built into the box's scratch copy and bench.py --timed-only (2000 updates at C3)
# is run on it; three rounds, alternating, so that drift of the box shows up as spread inside a variant rather than as a difference.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TESTS=0; if [ "$1" = "--tests" ]; then TESTS=1; shift; fi
S=reflector_ekf_slam_amd/csrc
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=8 -shared"
NAMES=()
while [ $# -gt 1 ]; do
  N=$1; F=$2; shift 2
  hipcc $BASE $F $S/ekf_kernels.hip $S/rekf_api.hip -o /tmp/librekf_$N.so || { echo "build of $N failed"; continue; }
  NAMES+=($N)
done
for R in 1 2 3; do
  for N in "${NAMES[@]}"; do
    cp /tmp/librekf_$N.so reflector_ekf_slam_amd/librekf.so
    V=$(python bench.py --steps 2000 --warmup 100 --timed-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f updates/s  %.3f us' % (d['value'], 1000*d['ms_per_step']))")
    echo "round $R  $N: $V" | tee -a gpurun_out/rekf_ab.txt
  done
done
if [ $TESTS = 1 ]; then
  for N in "${NAMES[@]}"; do
    cp /tmp/librekf_$N.so reflector_ekf_slam_amd/librekf.so
    echo "== tests on $N"; python -m pytest tests -x -q -m gpu -k "ekf or rekf or round or node or bench" 2>&1 | tail -3
  done
fi
