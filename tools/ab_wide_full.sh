#!/bin/bash
# Full-size A/B of the leaf-exact wide walk (HPRT_WIDE_WALK=1, the default) against the binary walk (=0): one warm-up step, then
# plain steps alternating between the two.   usage: tools/ab_wide_full.sh "<workloads>" [repeats]
R=$GRAFT_REPO_ROOT; cd $R
W=${1:-"atrium living-room"}; N=${2:-2}
for w in $W; do
  python3 bench.py --profile-step --workload $w > /dev/null 2>&1
  for i in $(seq 1 $N); do
    for v in 0 1; do echo -n "$w wide=$v "; HPRT_WIDE_WALK=$v python3 bench.py --profile-step --workload $w 2>/dev/null | grep profile_step; done
  done
done
