#!/bin/bash
# usage (GPU box, repo root): bash scripts/gpu_quick.sh <tag> [pytest args...]
# EKF parity tests, then a short bench: headline, per-kernel hipEvent brackets (each includes the ~4.7 us bracket), not_full.
TAG=${1:-q}; shift
timeout 900 python -m pytest ${@:-tests/test_ekf_round3_gpu.py tests/test_ekf_gpu.py} -m gpu -q -x 2>&1 | tail -6
python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --secondary "" --multi-sessions 0 --latency-steps 0 2>/dev/null | tail -1 > gpurun_out/bench_$TAG.json
python - <<PY
import json
d = json.load(open("gpurun_out/bench_$TAG.json"))
print("value", round(d["value"], 1), "us/step", round(1e3 * d["ms_per_step"], 2), "dd_us", round(d["roofline"]["avg_launch_us"], 2),
      "5pred", round(d["with_5_predicts_per_scan"]["value"], 1), "not_full", round(d["not_full"].get("value", 0), 1))
print(d["kernel_us"])
PY
