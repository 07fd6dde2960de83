#!/bin/bash
# usage (on the GPU box, from repo root): bash scripts/gpu_check_and_profile.sh <tag> [steps]
TAG=${1:-x}; STEPS=${2:-500}
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o $TAG -- python bench.py --steps $STEPS --warmup 50 --no-cpu-baseline > gpurun_out/bench_prof_$TAG.log 2>&1
tail -1 gpurun_out/bench_prof_$TAG.log | cut -c1-200
python bench.py --steps $STEPS --warmup 50 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_$TAG.json
python -c "
import json; d=json.load(open('gpurun_out/bench_$TAG.json')); print('value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'roofline frac', round(d['roofline']['frac'],3), 'mfma frac', round(d['roofline']['mfma']['frac'],3), d['kernel_us'])"
