#!/bin/bash
# round 6, first GPU pass: the new tests, the suites they touch, then the bench line (default run and the driver's command)
mkdir -p gpurun_out
python -m pytest tests/test_ekf_round6_gpu.py tests/test_ekf_round5_gpu.py tests/test_ekf_round4_gpu.py tests/test_ekf_round3_gpu.py -x -q -m gpu > gpurun_out/r6a_tests.txt 2>&1
tail -15 gpurun_out/r6a_tests.txt
python bench.py --steps 2000 > gpurun_out/r6a_bench_default.json 2> gpurun_out/r6a_bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6a_bench_driver.json 2> gpurun_out/r6a_bench_driver.err
python - <<'PY'
import json
for f in ("gpurun_out/r6a_bench_default.json", "gpurun_out/r6a_bench_driver.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value", round(d["value"]), "not_full", d.get("not_full", {}).get("value"), "fixed", d.get("fixed_capacity", {}).get("value"),
              "multi", d.get("multi_session", {}).get("value"), "readback", d.get("not_full", {}).get("with_pose_readback", {}).get("value"),
              "C2", d.get("secondary", {}).get("C2", {}).get("value"), "C4", d.get("secondary", {}).get("C4", {}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY
