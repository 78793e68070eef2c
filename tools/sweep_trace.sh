#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/sweep
for t in "52,24,6" "52,24,4" "52,24,3" "52,24,2" "52,64,5" "52,16,5" "58,24,5" "46,24,5" "40,24,5" "58,24,3" "52,32,4" "60,64,4" "32,24,5"; do
  echo "== tune $t"
  HPRT_TRACE_TUNE=$t timeout -k 10 200 python tools/bench_trace.py 8 2>/dev/null | grep Mrays
done
