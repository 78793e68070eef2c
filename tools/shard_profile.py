"""Renders rank 0's shard of an N-GPU run (tiles 0, N, 2N, ...) a few times, for rocprofv3 --kernel-trace --stats:
python3 tools/shard_profile.py <workload> <N> [renders]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hprt = importlib.import_module("thesis-pbrt-v3_amd")
import bench
name, n = sys.argv[1], int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
model = bench.build_model(hprt, name)
scene = hprt.Scene(model, hprt.Bvh(model))
opt = model.options.copy(); opt.spp = bench.WORKLOADS[name][1]
for _ in range(reps):
    t0 = time.perf_counter()
    _, st = scene.render(opt, tile_begin=0, tile_stride=n, export_foreign=n > 1)
    print("%s N=%d: %.2f ms" % (name, n, 1e3 * (time.perf_counter() - t0)), {k: st[k] for k in ("extend_launches", "extend_seconds", "occluded_seconds") if k in st}, flush=True)
