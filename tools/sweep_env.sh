#!/bin/bash
# usage: bash tools/sweep_env.sh VAR v1 v2 ...   (full-frame bench per value of an environment knob, with per-kernel times)
R=$GRAFT_REPO_ROOT; cd $R
make -s -C oracle
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py -m gpu -x -q 2>&1 | tail -3 || exit 1
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']
        print('$VAR=$v: %.1f ms/frame  %.0f Mrays/s  closest %.0f  any %.0f' % (j['ms_per_step'], j['value'], r['kernel_mrays_per_s'], r['occluded_kernel_mrays_per_s']))
"
done
