#!/bin/bash
# After tools/final_profiles.sh + tools/collect_profiles.py (so that the committed counter summary carries this build's stamp): the bench
# command under rocprofv3 (kernel stats + trace of its own launches) and the full bench line.   usage: bash tools/final_bench.sh <label>
L=${1:-final}; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/$L
make -s -C oracle
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$L/bench_command -- python3 $R/bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > $R/gpurun_out/$L/bench_command.log 2>&1; echo "bench command rc=$?")
timeout -k 10 900 python3 bench.py > gpurun_out/$L/final_bench.log 2>&1; echo "bench rc=$?"
grep '^{' gpurun_out/$L/final_bench.log | tail -1 > gpurun_out/$L/final_bench.json
grep '^{' gpurun_out/$L/bench_command.log | tail -1 > gpurun_out/$L/bench_command_line.json
python3 -c "
import json
j=json.load(open('gpurun_out/$L/final_bench.json'))
r=j['roofline']
print('headline', j['ms_per_step'], 'ms', j['value'], 'Mrays/s', r['kernel'], r.get('bound'), r.get('bound_candidates'), 'stale', r.get('counters_stale'), 'frac', r.get('frac'), 'avg_launch_ms', r.get('avg_launch_ms'), 'rocprof', r.get('rocprof_avg_launch_ms'))
for s in j['secondary']: print(s['workload'][:40], s['ms_per_step'], s['mrays_per_s'], s.get('gpu_over_cpu'), s['roofline'].get('bound_candidates'))
print('cpu', j['cpu_baseline']['value'] if j.get('cpu_baseline') else None, j.get('gpu_over_cpu'))
"
