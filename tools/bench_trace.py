"""Micro-benchmark of k_trace on HBM-resident rays (hprt_intersect_device / hprt_occluded_device).
Ray sets: 'primary' (camera rays, tile order) and 'bounce' (random directions leaving the
primary hit points, compacted — what bounce rays look like).  Prints Mrays/s per set."""
import importlib, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
hprt = importlib.import_module("thesis-pbrt-v3_amd")
import orc
FIX = os.path.join(ROOT, "tests", "golden", "killeroo_simple.hprt")
model = hprt.Model.load(FIX); bvh = hprt.Bvh(model); scene = hprt.Scene(model, bvh, device=0)
oracle = orc.OracleScene(FIX)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
# primary rays: all 1936 tiles, tile order, samples 0..reps-1
px = []; py = []
for ty in range(44):
    for tx in range(44):
        xs = np.arange(tx * 16, min(tx * 16 + 16, 700)); ys = np.arange(ty * 16, min(ty * 16 + 16, 700))
        X, Y = np.meshgrid(xs, ys); px.append(X.ravel()); py.append(Y.ravel())
px = np.concatenate(px).astype(np.int32); py = np.concatenate(py).astype(np.int32)
O = []; D = []
for s in range(reps):
    o, d = oracle.camera_rays(px, py, np.full(px.shape, s, np.int64)); O.append(o); D.append(d)
o = np.concatenate(O); d = np.concatenate(D); n = o.shape[0]
dev = torch.device("cuda", 0)
def to7(o, d, tmax):
    return torch.from_numpy(np.concatenate([o.T.ravel(), d.T.ravel(), tmax]).astype(np.float32)).to(dev)
def run(name, rays7, n, anyhit=False, iters=5):
    t = torch.empty(n, dtype=torch.float32, device=dev); prim = torch.empty(n, dtype=torch.int32, device=dev)
    occ = torch.empty(n, dtype=torch.uint8, device=dev)
    def go():
        if anyhit: scene.occluded_device(n, rays7.data_ptr(), occ.data_ptr())
        else: scene.intersect_device(n, rays7.data_ptr(), t.data_ptr(), prim.data_ptr())
    go(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): go()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print("%-22s n=%9d  %7.2f ms  %8.1f Mrays/s" % (name, n, dt * 1e3, n / dt / 1e6), flush=True)
    return t, prim
r7 = to7(o, d, np.full(n, np.inf, np.float32))
t, prim = run("primary closest", r7, n)
tt = t.cpu().numpy(); pp = prim.cpu().numpy(); hit = pp >= 0
rng = np.random.default_rng(0)
p = (o[hit] + d[hit] * tt[hit, None] * np.float32(0.9999)).astype(np.float32)
v = rng.normal(size=p.shape).astype(np.float32); v /= np.linalg.norm(v, axis=1, keepdims=True)
m = p.shape[0]
b7 = to7(p, v, np.full(m, np.inf, np.float32))
run("bounce closest", b7, m)
light = np.array([150.0, 120.0, 20.0], np.float32); to_o = p - light
target = light + to_o / np.linalg.norm(to_o, axis=1, keepdims=True) * np.float32(3.01)
s7 = to7(p, (target - p).astype(np.float32), np.full(m, np.float32(1) - np.float32(1e-4), np.float32))
run("shadow any-hit", s7, m, anyhit=True)
# ---- would reordering bounce rays by direction octant inside blocks of B consecutive rays pay? (diagnostics) ----
octant = ((v[:, 0] < 0).astype(np.int64) | ((v[:, 1] < 0).astype(np.int64) << 1) | ((v[:, 2] < 0).astype(np.int64) << 2))
for B in (512, 4096, 65536, m):
    block = np.arange(m, dtype=np.int64) // B
    order = np.lexsort((np.arange(m), octant, block))
    run("bounce octant/%d" % B, to7(p[order], v[order], np.full(m, np.inf, np.float32)), m)
# 6-bit key: octant + dominant axis
ax = np.argmax(np.abs(v), axis=1).astype(np.int64)
for B in (4096, 65536):
    block = np.arange(m, dtype=np.int64) // B
    order = np.lexsort((np.arange(m), octant * 3 + ax, block))
    run("bounce oct+axis/%d" % B, to7(p[order], v[order], np.full(m, np.inf, np.float32)), m)
