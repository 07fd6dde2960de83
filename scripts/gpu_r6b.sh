#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_ekf_round6_gpu.py tests/test_ekf_gpu.py tests/test_ekf_round3_gpu.py -x -q -m gpu > gpurun_out/r6b_tests.txt 2>&1
tail -15 gpurun_out/r6b_tests.txt
( python scripts/gpu_ab_multi.py 600; REKF_SCAN_LAUNCH=0 python scripts/gpu_ab_multi.py 600; REKF_SPEC=0 python scripts/gpu_ab_multi.py 600 ) > gpurun_out/r6b_multi.txt 2>&1
cat gpurun_out/r6b_multi.txt
