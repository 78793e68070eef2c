#!/bin/bash
# Frame time + trace-kernel rates of the three bench workloads (one plain step each after a warm-up step).  Usage: bash tools/quick_perf.sh
R=$GRAFT_REPO_ROOT; cd $R
for w in atrium killeroo-simple living-room; do
  python3 bench.py --profile-step --workload $w > /dev/null 2>&1   # warm-up (page-in, first-touch allocations)
  for i in 1 2; do python3 bench.py --profile-step --workload $w 2>/dev/null | grep profile_step; done
done
