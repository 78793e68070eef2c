#!/bin/bash
# usage: bash tools/sweep_atrium_tune.sh <out file> <closest|any> <scene> tune1 tune2 ...   (kernel rates of a tools/bench_scene.py scene at 256 spp per HPRT_TRACE_TUNE[_ANY] value)
OUT=$1; WHICH=$2; SCENE=$3; shift 3
VAR=HPRT_TRACE_TUNE; [ "$WHICH" = any ] && VAR=HPRT_TRACE_TUNE_ANY
for t in "$@"; do
  env $VAR=$t timeout -k 10 200 python tools/bench_scene.py $SCENE --spp 256 --steps 2 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$VAR=$t: %.1f ms closest %.0f any %.0f' % (j['ms_per_frame'], j['closest_kernel_mrays_per_s'], j['any_hit_kernel_mrays_per_s']))
" >> $OUT
done
cat $OUT
