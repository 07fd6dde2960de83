cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python scripts/gpu_bench_detectors.py 200 2>&1 | tail -8 | tee gpurun_out/detectors_r01q.jsonl
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_det -o det -- python scripts/gpu_bench_detectors.py 50 > /dev/null 2>&1
ls gpurun_out/prof_det/
