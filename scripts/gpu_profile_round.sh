#!/bin/bash
# usage (on the GPU box, from the repo root): bash scripts/gpu_profile_round.sh <tag> <commit> [steps]
# 1. rocprofv3 --kernel-trace --stats over bench.py --timed-only (per-kernel durations of the timed loop: ONE launch per update,
#    k_mid<4, 0> = mid role + the previous scan's downdate role + the next scan's speculative front end);
# 2./3. two SEPARATE --pmc passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only, as the MI355X guide prescribes;
# 4. a third --pmc pass: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (counter-based MFMA utilisation);
# 5. the detectors: call latencies, --kernel-trace --stats, the two --pmc passes (scripts/gpu_bench_detectors.py);
# 6. the bench lines (driver command and default) of the same tree.
# Summaries are written HERE (the databases are too big to travel) into gpurun_out/profiles_<tag>/; the builder runs
# `python scripts/profiles_commit.py <tag> <commit>` to move them into profiles/ and rewrite profiles/MANIFEST.json.
TAG=${1:-x}; COMMIT=${2:-unknown}; STEPS=${3:-1000}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/profiles_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o $TAG -- python bench.py --steps $STEPS --warmup 100 --timed-only > gpurun_out/bench_prof_$TAG.log 2>&1
tail -1 gpurun_out/bench_prof_$TAG.log | cut -c1-160
python scripts/rocpd_stats.py gpurun_out/prof_$TAG/${TAG}_results.db --last $STEPS --json $OUT/kernel_avg_us.json --commit $COMMIT > $OUT/${TAG}_kernel_stats.txt
head -8 $OUT/${TAG}_kernel_stats.txt | cut -c1-150
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_${TAG}_fetch -o fetch -- python bench.py --steps 200 --warmup 20 --timed-only > gpurun_out/bench_pmc_fetch_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_${TAG}_write -o write -- python bench.py --steps 200 --warmup 20 --timed-only > gpurun_out/bench_pmc_write_$TAG.log 2>&1
python scripts/pmc_summary.py gpurun_out/prof_${TAG}_fetch/fetch_results.db gpurun_out/prof_${TAG}_write/write_results.db $TAG 150 $OUT $COMMIT | tail -12
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/prof_${TAG}_mfma -o mfma -- python bench.py --steps 200 --warmup 20 --timed-only > gpurun_out/bench_pmc_mfma_$TAG.log 2>&1
python scripts/pmc_mfma.py gpurun_out/prof_${TAG}_mfma/mfma_results.db 150 --json $OUT/pmc_mfma.json --commit $COMMIT --tag $TAG > $OUT/${TAG}_pmc_mfma.txt 2>&1; grep -i "k_mid<4, 0>" $OUT/${TAG}_pmc_mfma.txt | cut -c1-200
rm -rf gpurun_out/prof_${TAG} gpurun_out/prof_${TAG}_fetch gpurun_out/prof_${TAG}_write gpurun_out/prof_${TAG}_mfma
# ---- detectors
timeout 600 python scripts/gpu_bench_detectors.py 200 2>&1 | tail -8 > $OUT/${TAG}_detectors.jsonl
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_det -o det -- python scripts/gpu_bench_detectors.py 50 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_det_fetch -o fetch -- python scripts/gpu_bench_detectors.py 20 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_det_write -o write -- python scripts/gpu_bench_detectors.py 20 > /dev/null 2>&1
python scripts/rocpd_stats.py gpurun_out/prof_det/det_results.db > $OUT/${TAG}_detectors_kernel_stats.txt 2>&1
python scripts/pmc_detectors.py gpurun_out/prof_det_fetch/fetch_results.db gpurun_out/prof_det_write/write_results.db > $OUT/${TAG}_detectors_pmc.txt 2>&1
head -14 $OUT/${TAG}_detectors_kernel_stats.txt | cut -c1-130
rm -rf gpurun_out/prof_det gpurun_out/prof_det_fetch gpurun_out/prof_det_write
# ---- bench lines of the same tree
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${TAG}_bench_driver_cmd.json
python bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench_default.json
cut -c1-260 $OUT/${TAG}_bench_driver_cmd.json; echo; cut -c1-260 $OUT/${TAG}_bench_default.json; echo
ls -la $OUT
