for rep in 1 2; do
for v in 0 1; do
REKF_ONE_LAUNCH=$v python bench.py --steps 200 --warmup 20 --no-cpu-baseline --multi-sessions 0 --secondary C2 --latency-steps 0 --detector-reps 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
c = d['secondary']['C2']
print('ONE_LAUNCH=$v  C2 %.0f updates/s (%.2f us)  5pred+readback %.0f  kernel_us %s   C3 %.0f' % (c['value'], c['us_per_update'], c['with_5_predicts_per_scan']['value'], c['kernel_us'], d['value']))
"
done; done
