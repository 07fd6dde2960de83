#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of k_mid<4, 0> on the FIXED-capacity filter (ld = 2112) next to the wrapper default (ld = 4160): do the counters
# or the kernel move more on the doubled-capacity layout?  (two separate --pmc passes each, as the guide prescribes)
TAG=${1:-fx}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in fixed default; do
  FLAG=""; [ $mode = fixed ] && FLAG="--fixed-capacity"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/p_${mode}_f -o f -- python bench.py --steps 200 --warmup 20 --timed-only $FLAG > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/p_${mode}_w -o w -- python bench.py --steps 200 --warmup 20 --timed-only $FLAG > /dev/null 2>&1
  mkdir -p gpurun_out/pmc_$mode
  python scripts/pmc_summary.py gpurun_out/p_${mode}_f/f_results.db gpurun_out/p_${mode}_w/w_results.db $TAG 150 gpurun_out/pmc_$mode none | grep "k_mid<4, 0>" | head -1 | sed "s/^/$mode: /"
  rm -rf gpurun_out/p_${mode}_f gpurun_out/p_${mode}_w
done
