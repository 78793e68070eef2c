#!/bin/bash
# Schedule sweep of the wide-walk kernels on one workload: "refillBelow,parkLimit,stepLimit,-,primMin" for closest-hit (HPRT_WALK4_TUNE)
# and any-hit rays (HPRT_WALK4_TUNE_ANY).   usage: tools/sweep_walk4.sh <workload> "<closest tunes>" "<any tunes>"
R=$GRAFT_REPO_ROOT; cd $R
w=${1:-atrium}
python3 bench.py --profile-step --workload $w > /dev/null 2>&1
for t in $2; do echo -n "$w closest tune=$t "; HPRT_WALK4_TUNE=$t python3 bench.py --profile-step --workload $w 2>/dev/null | grep profile_step; done
for t in $3; do echo -n "$w any tune=$t "; HPRT_WALK4_TUNE_ANY=$t python3 bench.py --profile-step --workload $w 2>/dev/null | grep profile_step; done
