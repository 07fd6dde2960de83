# usage (GPU box): bash scripts/probe/prof_c2.sh [path/to/librekf.so ...]   -- rocprofv3 kernel durations of the C2 chain per build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
i=0
for so in "$@"; do
i=$((i+1)); rm -rf gpurun_out/pc2_$i
cp $so reflector_ekf_slam_amd/librekf.so      # (the GPU box works on a scratch copy of the tree)
rocprofv3 --kernel-trace --stats -d gpurun_out/pc2_$i -o r -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --multi-sessions 0 --secondary C2 --latency-steps 0 --detector-reps 0 > gpurun_out/pc2_$i.json 2>/dev/null
echo "== $so: $(python -c "import json;d=json.loads([l for l in open('gpurun_out/pc2_$i.json') if l.startswith('{')][0]);print('C2 %.2f us/update' % d['secondary']['C2']['us_per_update'])")"
python scripts/rocpd_stats.py gpurun_out/pc2_$i/r_results.db 2>&1 | grep -E "k_mid<2, 0|k_dd_front<32>|k_mid<2, 2, 32" | cut -c1-150
done
