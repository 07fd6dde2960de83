cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python scripts/gpu_bench_grid.py 100 2>&1 | tail -7 | tee gpurun_out/grid_r01u.jsonl
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_grid -o grid -- python scripts/gpu_bench_grid.py 30 > /dev/null 2>&1
ls gpurun_out/prof_grid/
