#!/bin/bash
# Sweep of the any-hit walk knobs (refill threshold x step limit) on the three bench workloads; round 2 took 40,24,10,4,3 from it.  usage: bash tools/sweep_tune3.sh
R=$GRAFT_REPO_ROOT; cd $R
for W in atrium living-room killeroo-simple; do
python3 bench.py --profile-step --workload $W > /dev/null 2>&1
run() { echo -n "$W $1=$2: "; env $1=$2 python3 bench.py --profile-step --workload $W 2>/dev/null | grep profile_step | cut -c28-140; }
for v in 52,24,10,4,3 40,24,10,4,3 32,24,10,4,3 24,24,10,4,3 16,24,10,4,3 40,24,16,4,3 32,24,16,4,3 32,24,24,4,3 24,24,16,4,3; do run HPRT_TRACE_TUNE_ANY $v; done
done
