#!/bin/bash
# usage (GPU box, repo root): bash scripts/gpu_driver_line.sh [reps]
# The DRIVER's exact bench command (--gpus 1 --steps 20 --warmup 5), several times, value and ms_per_step of each: what BENCH_rNN.json
# will say, as opposed to the 2000-step figure the docs quote (a 20-step window pays its pipeline fill and the closing sync 20 times harder).
REPS=${1:-3}
for i in $(seq $REPS); do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --secondary "" --multi-sessions 0 --latency-steps 0 --detector-reps 0 --instr-steps 50 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps20 value %.0f us/step %.2f' % (d['value'], 1e3*d['ms_per_step']))"
done
python bench.py --steps 2000 --warmup 100 --no-cpu-baseline --secondary "" --multi-sessions 0 --latency-steps 0 --detector-reps 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps2000 value %.0f us/step %.2f' % (d['value'], 1e3*d['ms_per_step']))"
