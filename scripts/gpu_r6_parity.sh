#!/bin/bash
# long parity runs on the final tree (output -> gpurun_out/r6_parity.txt)
mkdir -p gpurun_out
O=gpurun_out/r6_parity.txt
: > $O
echo "== C3 soak 3000" >> $O; timeout 900 python scripts/gpu_long_parity_c3.py 3000 2>&1 | tail -2 >> $O
echo "== fuzz default 400 (first 11000)" >> $O; timeout 900 python scripts/gpu_fuzz_ekf.py 400 11000 2>&1 | tail -2 >> $O
echo "== fuzz burst 400 (first 12000)" >> $O; timeout 900 python scripts/gpu_fuzz_ekf.py 400 12000 burst 2>&1 | tail -2 >> $O
echo "== fuzz steady 600 (first 13000)" >> $O; timeout 1200 python scripts/gpu_fuzz_ekf.py 600 13000 steady 2>&1 | tail -2 >> $O
echo "== detectors fuzz" >> $O; timeout 600 python scripts/gpu_fuzz_detectors.py 20 300 2>&1 | tail -2 >> $O
cat $O
