#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
make -s -C oracle
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py -m gpu -x -q 2>&1 | tail -2
for t in "52,24,6,1" "52,24,6,8" "52,24,6,16" "52,24,6,24" "52,24,6,32" "52,24,6,48"; do
  echo "== tune $t"
  HPRT_TRACE_TUNE=$t timeout -k 10 200 python tools/bench_trace.py 8 2>/dev/null | grep Mrays
done
