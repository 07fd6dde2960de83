for SEC in "" C2 C4; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --latency-steps 0 --instr-steps 0 --secondary "$SEC" --secondary-steps 200 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('secondary [$SEC]', json.dumps(d['detectors']['cloud_3d']['call_us']), json.dumps(d['detectors']['laser_2d']['call_us']))"
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary "" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('with latency+instr legs', json.dumps(d['detectors']['cloud_3d']['call_us']))"
python bench.py --steps 20 --warmup 5 --latency-steps 0 --instr-steps 0 --secondary "" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('with cpu baseline', json.dumps(d['detectors']['cloud_3d']['call_us']))"
