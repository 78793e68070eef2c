#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
make -s -C oracle
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py -m gpu -x -q 2>&1 | tail -15 || exit 1
for t in "52,24,6,16,1" "52,24,6,16,8" "52,24,6,16,16" "52,24,6,16,24" "52,32,6,16,24" "52,32,6,16,32" "52,40,8,16,32" "56,40,8,16,40"; do
  echo "== tune $t"
  HPRT_TRACE_TUNE=$t timeout -k 10 200 python tools/bench_trace.py 8 2>/dev/null | grep Mrays
done
HPRT_TRACE_TUNE=52,32,6,16,24 timeout -k 10 300 python tools/trace_profile.py 128 2>/dev/null
