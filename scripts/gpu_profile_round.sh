#!/bin/bash
# usage (on the GPU box, from the repo root): bash scripts/gpu_profile_round.sh <tag> <commit> [steps]
# 1. rocprofv3 --kernel-trace --stats over bench.py --timed-only (per-kernel durations of the timed loop: since round 5 ONE launch per update,
#    k_mid<4, 0> = mid role + the previous scan's downdate role + the next scan's speculative front end);
# 2./3. two SEPARATE --pmc passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only, as the MI355X guide prescribes;
# 4. a third --pmc pass: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (counter-based MFMA utilisation);
# 5. the bench lines (driver command and default) of the same tree.
# Summaries are written HERE (the databases are too big to travel) into profiles/ -> gpurun_out/profiles_<tag>/ for the builder to commit.
TAG=${1:-x}; COMMIT=${2:-unknown}; STEPS=${3:-1000}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/profiles_$TAG; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o $TAG -- python bench.py --steps $STEPS --warmup 100 --timed-only > gpurun_out/bench_prof_$TAG.log 2>&1
tail -1 gpurun_out/bench_prof_$TAG.log | cut -c1-160
python scripts/rocpd_stats.py gpurun_out/prof_$TAG/${TAG}_results.db --last $STEPS --json $OUT/kernel_avg_us.json --commit $COMMIT > $OUT/${TAG}_kernel_stats.txt
head -8 $OUT/${TAG}_kernel_stats.txt | cut -c1-150
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_${TAG}_fetch -o fetch -- python bench.py --steps 200 --warmup 20 --timed-only > gpurun_out/bench_pmc_fetch_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_${TAG}_write -o write -- python bench.py --steps 200 --warmup 20 --timed-only > gpurun_out/bench_pmc_write_$TAG.log 2>&1
python scripts/pmc_summary.py gpurun_out/prof_${TAG}_fetch/fetch_results.db gpurun_out/prof_${TAG}_write/write_results.db $TAG 150 $OUT $COMMIT | tail -12
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/prof_${TAG}_mfma -o mfma -- python bench.py --steps 200 --warmup 20 --timed-only > gpurun_out/bench_pmc_mfma_$TAG.log 2>&1
python scripts/pmc_mfma.py gpurun_out/prof_${TAG}_mfma/mfma_results.db 150 > $OUT/${TAG}_pmc_mfma.txt 2>&1; grep -i "k_mid<4, 0>\|k_downdate2<64" $OUT/${TAG}_pmc_mfma.txt | cut -c1-200
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${TAG}_bench_driver_cmd.json
python bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench_default.json
cut -c1-260 $OUT/${TAG}_bench_driver_cmd.json; echo; cut -c1-260 $OUT/${TAG}_bench_default.json; echo
rm -rf gpurun_out/prof_${TAG} gpurun_out/prof_${TAG}_fetch gpurun_out/prof_${TAG}_write gpurun_out/prof_${TAG}_mfma
ls -la $OUT
