#!/bin/bash
# Sweep the scheduling knobs of the persistent walk on one workload (plain steps).  usage: bash tools/sweep_tune.sh <workload> <VAR> v1 v2 ...
R=$GRAFT_REPO_ROOT; cd $R
W=$1; VAR=$2; shift 2
python3 bench.py --profile-step --workload $W > /dev/null 2>&1
for v in "$@"; do
  echo -n "$VAR=$v: "; env $VAR=$v python3 bench.py --profile-step --workload $W 2>/dev/null | grep profile_step | cut -c1-140
done
