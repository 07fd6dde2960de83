cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 0 1; do
rm -rf gpurun_out/pc2_$v
REKF_ONE_LAUNCH=$v rocprofv3 --kernel-trace --stats -d gpurun_out/pc2_$v -o r -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --multi-sessions 0 --secondary C2 --latency-steps 0 --detector-reps 0 > /dev/null 2>&1
echo "== ONE_LAUNCH=$v"; python scripts/rocpd_stats.py gpurun_out/pc2_$v/r_results.db 2>&1 | grep -E "Li2E|ILi32E|k_front|Name|name" | cut -c1-200 | head -12
done
