#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r6_fuzz_more.txt
: > $O
for spec in "1500 20000" "1500 30000 burst" "1500 40000 steady"; do
  echo "== fuzz $spec" >> $O
  timeout 1500 python scripts/gpu_fuzz_ekf.py $spec 2>&1 | grep -v '"ok": true' | cut -c1-420 >> $O
done
cat $O
