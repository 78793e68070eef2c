#!/bin/bash
# A/B frame times of library variants on one box (tools/build_variant.sh): for each workload, one warm-up step and two plain
# steps per variant, variants interleaved.   usage: tools/ab_perf.sh "<variant suffixes, '-' = the shipped library>" [workloads]
R=$GRAFT_REPO_ROOT; cd $R
V=${1:-"- a"}; W=${2:-"atrium killeroo-simple living-room"}
for w in $W; do
  for v in $V; do
    if [ "$v" = "-" ]; then L=$R/thesis-pbrt-v3_amd/lib/libhprt.so; else L=$R/thesis-pbrt-v3_amd/lib/libhprt_$v.so; fi
    HPRT_LIB=$L python3 bench.py --profile-step --workload $w > /dev/null 2>&1
    for i in 1 2; do echo -n "$w [$v] "; HPRT_LIB=$L python3 bench.py --profile-step --workload $w 2>/dev/null | grep profile_step; done
  done
done
