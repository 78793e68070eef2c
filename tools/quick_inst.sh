#!/bin/bash
# Frame times of the instanced 10M-triangle scene (tools/bench_scene.py) and killeroo-simple: the kernels WITH quadric / instance code.
R=$GRAFT_REPO_ROOT; cd $R
python3 tools/bench_scene.py instanced --spp 256 --steps 3 2>/dev/null | cut -c1-600
python3 bench.py --profile-step --workload killeroo-simple > /dev/null 2>&1
for i in 1 2 3; do python3 bench.py --profile-step --workload killeroo-simple 2>/dev/null | grep profile_step; done
