#!/bin/bash
# Plain steps of one workload per value of an environment knob.   usage: tools/sweep_step_env.sh <workload> VAR v1 v2 ...
R=$GRAFT_REPO_ROOT; cd $R
w=$1; VAR=$2; shift; shift
python3 bench.py --profile-step --workload $w > /dev/null 2>&1
for v in "$@"; do echo -n "$w $VAR=$v "; env $VAR=$v python3 bench.py --profile-step --workload $w 2>/dev/null | grep profile_step; done
