#!/bin/bash
# Instancing check on the GPU box: parity tests, then the config-5 stand-in's frame rate and phase profile.
R=$GRAFT_REPO_ROOT; cd $R
make -s -C oracle
timeout -k 10 400 python -m pytest tests/test_gpu_scenes.py tests/test_gpu_parity.py tests/test_gpu_textures.py -m gpu -x -q 2>&1 | tail -3 || exit 1
timeout -k 10 300 python tools/bench_scene.py instanced --spp ${1:-64} --steps 2 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('instanced: %.1f ms  %.0f Mrays/s  closest %.0f any %.0f' % (j['ms_per_frame'], j['mrays_per_s'], j['closest_kernel_mrays_per_s'], j['any_hit_kernel_mrays_per_s']))
" && timeout -k 10 200 python tools/trace_profile.py 32 instanced 2>/dev/null
