#!/bin/bash
# usage (on the GPU box, from the repo root): bash scripts/gpu_profile_round.sh <tag> [steps]
# 1. rocprofv3 --kernel-trace --stats over bench.py (per-kernel durations);
# 2./3. two SEPARATE --pmc passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only, as the MI355X guide prescribes.
# (bench.py --timed-only: map build + warm-up + timed steps, so that "--last N" in the summaries selects the timed region.)
# Everything lands in gpurun_out/prof_<tag>*/ ; summarise locally with scripts/rocpd_stats.py (--last 1000) and scripts/pmc_summary.py.
TAG=${1:-x}; STEPS=${2:-1000}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o $TAG -- python bench.py --steps $STEPS --warmup 100 --timed-only > gpurun_out/bench_prof_$TAG.log 2>&1
tail -1 gpurun_out/bench_prof_$TAG.log | cut -c1-160
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_${TAG}_fetch -o fetch -- python bench.py --steps 200 --warmup 20 --timed-only > gpurun_out/bench_pmc_fetch_$TAG.log 2>&1
tail -1 gpurun_out/bench_pmc_fetch_$TAG.log | cut -c1-100
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_${TAG}_write -o write -- python bench.py --steps 200 --warmup 20 --timed-only > gpurun_out/bench_pmc_write_$TAG.log 2>&1
tail -1 gpurun_out/bench_pmc_write_$TAG.log | cut -c1-100
python bench.py --steps $STEPS --warmup 100 2>/dev/null | tail -1 > gpurun_out/bench_$TAG.json
cut -c1-300 gpurun_out/bench_$TAG.json
ls -la gpurun_out/prof_${TAG}*/
