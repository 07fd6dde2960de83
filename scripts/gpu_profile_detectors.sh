#!/bin/bash
# usage (GPU box, repo root): bash scripts/gpu_profile_detectors.sh [tag]
# 1. per-call latency of the 2D / 3D detectors next to the CPU oracle (gpu_bench_detectors.py);
# 2. rocprofv3 --kernel-trace --stats over the same script (per-kernel durations of k_det2d, k3_*);
# 3./4. two SEPARATE --pmc passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only, as the MI355X guide prescribes.
# Everything lands in gpurun_out/; summarise with scripts/rocpd_stats.py and scripts/pmc_detectors.py into profiles/.
TAG=${1:-det}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python scripts/gpu_bench_detectors.py 200 2>&1 | tail -8 | tee gpurun_out/detectors_$TAG.jsonl
rm -rf gpurun_out/prof_det gpurun_out/prof_det_fetch gpurun_out/prof_det_write
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_det -o det -- python scripts/gpu_bench_detectors.py 50 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_det_fetch -o fetch -- python scripts/gpu_bench_detectors.py 20 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_det_write -o write -- python scripts/gpu_bench_detectors.py 20 > /dev/null 2>&1
python scripts/rocpd_stats.py gpurun_out/prof_det/det_results.db > gpurun_out/${TAG}_kernel_stats.txt 2>&1
python scripts/pmc_detectors.py gpurun_out/prof_det_fetch/fetch_results.db gpurun_out/prof_det_write/write_results.db > gpurun_out/${TAG}_pmc.txt 2>&1
ls gpurun_out/prof_det/ gpurun_out/prof_det_fetch gpurun_out/prof_det_write
