#!/bin/bash
# One-at-a-time sweep of the walk's scheduling knobs around the defaults (closest hit: large-scene defaults 52,32,14,4,12; any hit 52,24,10,4,3).
# usage: bash tools/sweep_tune2.sh <workload>
R=$GRAFT_REPO_ROOT; cd $R; W=$1
python3 bench.py --profile-step --workload $W > /dev/null 2>&1
run() { echo -n "$1=$2: "; env $1=$2 python3 bench.py --profile-step --workload $W 2>/dev/null | grep profile_step | cut -c1-140; }
run HPRT_TRACE_TUNE 52,32,14,4,12
for v in 52,32,10,4,12 52,32,20,4,12 52,32,28,4,12 52,24,14,4,12 52,40,14,4,12 52,48,14,4,12 52,32,14,4,8 52,32,14,4,16 52,32,14,4,24 40,32,14,4,12 60,32,14,4,12; do run HPRT_TRACE_TUNE $v; done
run HPRT_TRACE_TUNE_ANY 52,24,10,4,3
for v in 52,24,10,4,1 52,24,10,4,6 52,16,10,4,3 52,32,10,4,3 52,24,6,4,3 52,24,16,4,3 40,24,10,4,3 60,24,10,4,3; do run HPRT_TRACE_TUNE_ANY $v; done
