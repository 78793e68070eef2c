"""A/B of the leaf-exact four-wide walk (k_walk4) against the binary walk (k_trace) in ONE process, same scene, same rays.
usage: python tools/ab_wide.py [atrium|living-room|killeroo] [--spp N] [--rays M]
1. random rays (axis-parallel directions, origins on node planes and finite segments among them) through hprt_intersect /
   hprt_occluded with either walk: t, primitive, barycentrics and flags must be bit-identical;
2. the frame at --spp with either walk: films bit-identical, frame times and kernel rates of both."""
import argparse, importlib, json, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
hprt = importlib.import_module("thesis-pbrt-v3_amd")
import ctypes as C

ap = argparse.ArgumentParser()
ap.add_argument("scene", nargs="?", default="atrium"); ap.add_argument("--spp", type=int, default=16); ap.add_argument("--rays", type=int, default=4_000_000)
ap.add_argument("--steps", type=int, default=2)
args = ap.parse_args()
lib = hprt.lib if hasattr(hprt, "lib") else None
if lib is None:
    import importlib as _i
    lib = sys.modules["thesis-pbrt-v3_amd"].lib
lib.hprt_debug_wide_walk.argtypes = [C.c_int]; lib.hprt_debug_wide_walk.restype = C.c_int
if args.scene == "atrium":
    import scene_gen
    text, _ = scene_gen.atrium(1.0)
    path = os.path.join(tempfile.mkdtemp(), "atrium.pbrt"); open(path, "w").write(text)
    model = hprt.Model.parse(path)
elif args.scene == "instanced-10m":
    import scene_gen
    text, _ = scene_gen.instanced_killeroo(os.path.join(ROOT, "tests", "golden", "killeroo.hprt"))
    path = os.path.join(tempfile.mkdtemp(), "instanced10m.pbrt"); open(path, "w").write(text)
    model = hprt.Model.parse(path)
elif args.scene == "killeroo-simple":
    model = hprt.Model.load(os.path.join(ROOT, "tests", "golden", "killeroo_simple.hprt"))
elif args.scene == "living-room":
    model = hprt.Model.load(os.path.join(ROOT, "tests", "golden", "living_room.hprt"))
else:
    model = hprt.Model.load(os.path.join(ROOT, "tests", "golden", "killeroo.hprt"))
bvh = hprt.Bvh(model)
scene = hprt.Scene(model, bvh, device=0)
nodes, _ = bvh.arrays()
lo = nodes[:, 0:3].view(np.float32); hi = nodes[:, 3:6].view(np.float32)
rng = np.random.default_rng(11)
n = args.rays
ext = hi[0] - lo[0]
o = (lo[0] + rng.uniform(-0.1, 1.1, (n, 3)) * ext).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32)
k = n // 8
d[:k, 0] = 0.0; d[k:2 * k, 1] = -0.0; d[2 * k:3 * k, [0, 2]] = 0.0                       # axis-parallel planes and lines
pick = rng.integers(0, nodes.shape[0], 2 * k); ax = rng.integers(0, 3, 2 * k)          # origins exactly on node planes
o[np.arange(2 * k), ax] = np.where(rng.integers(0, 2, 2 * k) == 1, hi[pick, ax], lo[pick, ax])
tmax = np.full(n, np.inf, np.float32); tmax[n // 2:] = rng.uniform(0, np.linalg.norm(ext), n - n // 2).astype(np.float32)
res = {}
for wide in (0, 1):
    lib.hprt_debug_wide_walk(wide)
    if args.scene == "instanced-10m":
        t, p, inst_, b = scene.intersect_instanced(o, d, tmax); b = np.concatenate([b, inst_[:, None].astype(np.float32)], axis=1)
    else:
        t, p, b = scene.intersect(o, d, tmax)
    occ = scene.occluded(o, d, tmax)
    res[wide] = (t.copy(), p.copy(), b.copy(), occ.copy())
same = all(np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b_).view(np.uint8)) for a, b_ in zip(res[0], res[1]))
hits = int((res[0][1] >= 0).sum()); occl = int(res[0][3].sum())
print("random rays: %d rays, %d hits, %d occluded: %s" % (n, hits, occl, "bit-identical" if same else "MISMATCH"), flush=True)
if not same:
    for name, a, b_ in zip(("t", "prim", "bary", "occ"), res[0], res[1]):
        bad = np.nonzero((np.ascontiguousarray(a).view(np.uint8).reshape(n, -1) != np.ascontiguousarray(b_).view(np.uint8).reshape(n, -1)).any(axis=1))[0]
        print(" ", name, bad.size, "differ; first", bad[:5], a[bad[:3]], b_[bad[:3]])
opt = model.options.copy(); opt.spp = args.spp
films = {}; out = {}
for wide in (0, 1, 0, 1):
    lib.hprt_debug_wide_walk(wide)
    scene.render(opt)
    best = None
    for _ in range(args.steps):
        t0 = time.perf_counter(); film, st = scene.render(opt); dt = time.perf_counter() - t0
        if best is None or dt < best[0]: best = (dt, st)
    films[wide] = film.copy()
    dt, st = best
    out[wide] = {"ms": round(dt * 1e3, 2), "closest_grays": round(st["extend_rays"] / max(st["extend_seconds"], 1e-12) / 1e9, 3),
                 "any_grays": round(st["occluded_rays"] / max(st["occluded_seconds"], 1e-12) / 1e9, 3)}
    print("wide=%d" % wide, json.dumps(out[wide]), flush=True)
ok = np.array_equal(films[0].view(np.uint32), films[1].view(np.uint32))
print("films at %d spp: %s" % (args.spp, "bit-identical" if ok else "MISMATCH (%d pixels)" % int((films[0] != films[1]).any(axis=-1).sum())))
sys.exit(0 if (same and ok) else 1)
