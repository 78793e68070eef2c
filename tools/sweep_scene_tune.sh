#!/bin/bash
# usage: bash tools/sweep_scene_tune.sh <scene> <spp> VAR v1 v2 ...   (frame rate of a tools/bench_scene.py scene per value of an env knob)
R=$GRAFT_REPO_ROOT; cd $R
SCENE=$1; SPP=$2; VAR=$3; shift 3
for v in "$@"; do
  env $VAR=$v timeout -k 10 200 python tools/bench_scene.py $SCENE --spp $SPP --steps 2 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$VAR=$v: %.1f ms  %.0f Mrays/s  closest %.0f any %.0f' % (j['ms_per_frame'], j['mrays_per_s'], j['closest_kernel_mrays_per_s'], j['any_hit_kernel_mrays_per_s']))
"
done
