#!/bin/bash
# The judged measurements of a build, in one box session: counter passes of the four bench workloads (atrium with the L1 / L2 passes),
# rocprofv3 kernel stats + trace of the bench command itself, and the full bench line.   usage: bash tools/final_profiles.sh <label>
# afterwards (build container): python tools/collect_profiles.py <label> r03; cp gpurun_out/<label>/final_bench.json profiles/r03_final_bench.json ...
L=${1:-final}; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/$L
make -s -C oracle
HPRT_L1_PASSES=1 bash tools/counters_passes.sh $L atrium > gpurun_out/$L/passes_atrium.log 2>&1; tail -c 300 gpurun_out/$L/passes_atrium.log; echo
for w in killeroo-simple living-room instanced-10m; do bash tools/counters_passes.sh $L $w > gpurun_out/$L/passes_$w.log 2>&1; grep -h "rc=" gpurun_out/$L/passes_$w.log | cut -c1-150; done
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$L/bench_command -- python3 $R/bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > $R/gpurun_out/$L/bench_command.log 2>&1; echo "bench command rc=$?")
timeout -k 10 900 python3 bench.py > gpurun_out/$L/final_bench.log 2>&1; echo "bench rc=$?"
grep '^{' gpurun_out/$L/final_bench.log | tail -1 > gpurun_out/$L/final_bench.json
grep '^{' gpurun_out/$L/bench_command.log | tail -1 > gpurun_out/$L/bench_command_line.json
python3 -c "
import json
j=json.load(open('gpurun_out/$L/final_bench.json'))
print('headline', j['ms_per_step'], 'ms', j['value'], 'Mrays/s', j['roofline']['kernel'], j['roofline'].get('bound'), j['roofline'].get('bound_candidates'), 'stale', j['roofline'].get('counters_stale'))
for s in j['secondary']: print(s['workload'][:40], s['ms_per_step'], s['mrays_per_s'], s.get('gpu_over_cpu'))
print('cpu', j['cpu_baseline']['value'] if j.get('cpu_baseline') else None, j.get('gpu_over_cpu'))
"
