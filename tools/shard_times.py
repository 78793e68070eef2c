"""What a rank of an N-GPU run costs, measured on ONE GPU: the time of rank r's shard (tiles r, r+N, ... of the 16x16 grid at the
frame's full spp, HPRT_RENDER_EXPORT_FOREIGN as in the N-GPU run) for N = 1, 2, 4, 8, against 1/N of the whole frame.  The gap
is the fixed cost per render (launch tails, per-bounce read-backs, film kernels) that strong scaling cannot divide.
usage: python3 tools/shard_times.py [workload ...]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hprt = importlib.import_module("thesis-pbrt-v3_amd")
import bench

for name in (sys.argv[1:] or ["atrium", "killeroo-simple"]):
    model = bench.build_model(hprt, name)
    scene = hprt.Scene(model, hprt.Bvh(model))
    opt = model.options.copy(); opt.spp = bench.WORKLOADS[name][1]
    scene.render(opt)                                             # allocations
    full = None
    for n in (1, 2, 4, 8):
        worst = 0.0
        for r in sorted({0, n // 2, n - 1}):
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                scene.render(opt, tile_begin=r, tile_stride=n, export_foreign=n > 1)
                best = min(best, time.perf_counter() - t0)
            worst = max(worst, best)
        if n == 1: full = worst
        print("%-16s N=%d: slowest of the sampled ranks %8.2f ms, whole frame / N %8.2f ms, efficiency bound %.3f" % (name, n, worst * 1e3, full / n * 1e3, full / n / worst), flush=True)
