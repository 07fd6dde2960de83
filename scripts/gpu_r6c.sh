#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_ekf_round6_gpu.py -x -q -m gpu > gpurun_out/r6c_tests.txt 2>&1
tail -25 gpurun_out/r6c_tests.txt
python -m pytest tests/test_ekf_gpu.py tests/test_ekf_round3_gpu.py tests/test_ekf_round4_gpu.py tests/test_ekf_round5_gpu.py -x -q -m gpu > gpurun_out/r6c_tests2.txt 2>&1
tail -8 gpurun_out/r6c_tests2.txt
python bench.py --steps 2000 --no-cpu-baseline --detector-reps 0 > gpurun_out/r6c_bench.json 2> gpurun_out/r6c_bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --detector-reps 0 --secondary "" --multi-sessions 0 > gpurun_out/r6c_bench_driver.json 2>> gpurun_out/r6c_bench.err
python - <<'PY'
import json
for f in ("gpurun_out/r6c_bench.json", "gpurun_out/r6c_bench_driver.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value", round(d["value"]), "not_full", d.get("not_full", {}).get("value"), "fixed", d.get("fixed_capacity", {}).get("value"),
              "multi", d.get("multi_session", {}).get("value"), "readback", d.get("not_full", {}).get("with_pose_readback", {}).get("value"),
              "5pred", d.get("with_5_predicts_per_scan", {}).get("value"))
        print("  latency", {k: (v.get("median") if isinstance(v, dict) else v) for k, v in d.get("latency_us", {}).items()})
        print("  C2", d.get("secondary", {}).get("C2", {}).get("value"), "C4", d.get("secondary", {}).get("C4", {}).get("value"), d.get("secondary", {}).get("C4", {}).get("detector_pipeline", {}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY
