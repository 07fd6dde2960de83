#!/bin/bash
# usage (GPU box): bash scripts/gpu_c4_pipeline.sh   -- the C4 cloud -> 3D detector -> filter pipeline of bench.py alone (synchronous call, and two clouds on their way)
for R in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --latency-steps 0 --instr-steps 0 --multi-sessions 1 --detector-reps 100 --secondary C4 --secondary-steps 300 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['secondary']['C4']['detector_pipeline']
print('C4 pipeline: synchronous %.0f scans/s (%.1f us)   two on their way %.0f scans/s (%.1f us) identical %s   3D call %.1f us' % (p['value'], p['us_per_scan'], p['overlapped']['value'], p['overlapped']['us_per_scan'], p['overlapped']['identical_observations'], d['detectors']['cloud_3d']['call_us']['median']))"
done
