"""Would a spatial ray sort pay?  Upper bound, measured without writing the sort.

Renders a few spp of a bench workload, captures the rays one bounce queues (hprt_debug_capture_rays: path segments entering
bounce b + 1, or bounce b's shadow rays), and replays them through hprt_intersect_device / hprt_occluded_device
  (1) in the order the render queues them,
  (2) shuffled (how much coherence that order already has),
  (3) sorted on the host by several keys (Morton cell of the origin over the scene bound at 2^bits cells per axis,
      optionally with the direction's octant / cube-map cell in front of or behind it).
The replay reads the rays physically in that order (no index gather), so (3) is the ceiling a device-side sort can reach
before its own cost.  Usage: sort_experiment.py <workload> <spp> [bounces...]"""
import importlib, os, sys, time, ctypes as C
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
hprt = importlib.import_module("thesis-pbrt-v3_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
bounces = [int(x) for x in sys.argv[3:]] or [0, 1, 2]
dev = torch.device("cuda", 0)
model = bench.build_model(hprt, name); bvh = hprt.Bvh(model); scene = hprt.Scene(model, bvh, device=0)
opt = model.options.copy(); opt.spp = spp
bounds = np.array(bvh.info()["bounds"], np.float32); lo, hi = bounds[:3], bounds[3:]
print("workload %s spp %d bounds %s" % (name, spp, bounds), flush=True)
lib = hprt.lib
lib.hprt_debug_capture_rays.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
lib.hprt_debug_captured.restype = C.c_longlong; lib.hprt_debug_captured.argtypes = [C.c_void_p]
x0, y0, x1, y1 = opt.film_bounds()
cap = (x1 - x0) * (y1 - y0) * spp


def capture(bounce, kind):
    buf = torch.zeros(7 * cap, dtype=torch.float32, device=dev)
    hprt._check(lib.hprt_debug_capture_rays(scene._h, bounce, kind, buf.data_ptr(), cap))
    scene.render(opt, spp_chunk=spp)
    n = int(lib.hprt_debug_captured(scene._h))
    hprt._check(lib.hprt_debug_capture_rays(scene._h, -1, 0, None, 0))
    r = buf.view(7, cap)[:, :n].cpu().numpy()
    del buf
    return r      # [7][n]


def run(label, r7, anyhit, iters=4):
    n = r7.shape[1]
    rays = torch.from_numpy(np.ascontiguousarray(r7)).to(dev)
    t = torch.empty(n, dtype=torch.float32, device=dev); prim = torch.empty(n, dtype=torch.int32, device=dev)
    occ = torch.empty(n, dtype=torch.uint8, device=dev)
    def go():
        if anyhit: scene.occluded_device(n, rays.data_ptr(), occ.data_ptr())
        else: scene.intersect_device(n, rays.data_ptr(), t.data_ptr(), prim.data_ptr())
    go(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): go()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print("  %-34s n=%9d %8.2f ms %8.1f Mrays/s" % (label, n, dt * 1e3, n / dt / 1e6), flush=True)
    return n / dt / 1e6


def part1by2(v):
    v = v.astype(np.uint64) & np.uint64(0x1fffff)
    v = (v | (v << np.uint64(32))) & np.uint64(0x1f00000000ffff)
    v = (v | (v << np.uint64(16))) & np.uint64(0x1f0000ff0000ff)
    v = (v | (v << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
    v = (v | (v << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
    v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
    return v


def morton(o, bits):
    q = np.clip(((o - lo[:, None]) / np.maximum(hi - lo, 1e-20)[:, None] * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    return part1by2(q[0]) | (part1by2(q[1]) << np.uint64(1)) | (part1by2(q[2]) << np.uint64(2))


def octant(d):
    return ((d[0] < 0).astype(np.uint64) | ((d[1] < 0).astype(np.uint64) << np.uint64(1)) | ((d[2] < 0).astype(np.uint64) << np.uint64(2)))


def cubecell(d, k):
    """direction -> cube-map face (6) x k x k cell"""
    a = np.abs(d); ax = np.argmax(a, axis=0)
    m = np.take_along_axis(a, ax[None], 0)[0]
    sgn = np.take_along_axis(d, ax[None], 0)[0] < 0
    u = np.take_along_axis(d, ((ax + 1) % 3)[None], 0)[0] / np.maximum(m, 1e-30)
    v = np.take_along_axis(d, ((ax + 2) % 3)[None], 0)[0] / np.maximum(m, 1e-30)
    iu = np.clip(((u * 0.5 + 0.5) * k).astype(np.int64), 0, k - 1); iv = np.clip(((v * 0.5 + 0.5) * k).astype(np.int64), 0, k - 1)
    return ((ax.astype(np.int64) * 2 + sgn) * k * k + iu * k + iv).astype(np.uint64)


lib.hprt_debug_trace_queued.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_float)]


def run_queued(label, r7, order, anyhit, iters=4):
    """the same rays left in render order, consumed through the sorted index queue: what a device-side INDEX sort would give"""
    n = r7.shape[1]
    rays = torch.from_numpy(np.ascontiguousarray(r7)).to(dev); q = torch.from_numpy(order.astype(np.uint32).view(np.int32)).to(dev)
    scratch = torch.empty(7 * n, dtype=torch.float32, device=dev)
    ms = (C.c_float * 2)(); best = [1e9, 1e9]
    for _ in range(iters + 1):
        hprt._check(lib.hprt_debug_trace_queued(scene._h, n, rays.data_ptr(), q.data_ptr(), 1 if anyhit else 0, scratch.data_ptr(), ms))
        best = [min(best[0], ms[0]), min(best[1], ms[1])]
    print("  %-34s n=%9d %8.2f ms %8.1f Mrays/s   (+ permutation pass alone: %.2f ms)" % (label + " [queue]", n, best[1], n / best[1] / 1e3, best[0]), flush=True)


rng = np.random.default_rng(1)
for b in bounces:
    for kind, anyhit, what in ((0, False, "path rays entering bounce %d" % (b + 1)), (1, True, "shadow rays of bounce %d" % b)):
        r = capture(b, kind)
        n = r.shape[1]
        if n == 0: continue
        print("== %s: %d rays" % (what, n), flush=True)
        o, d = r[0:3], r[3:6]
        base = run("render order", r, anyhit)
        run("shuffled", r[:, rng.permutation(n)], anyhit)
        keys = {}
        for bits in (4, 6, 8, 10):
            keys["morton%d" % bits] = morton(o, bits)
        keys["oct|morton6"] = (octant(d) << np.uint64(18)) | morton(o, 6)
        keys["morton6|oct"] = (morton(o, 6) << np.uint64(3)) | octant(d)
        keys["morton8|oct"] = (morton(o, 8) << np.uint64(3)) | octant(d)
        keys["morton5|cube4"] = (morton(o, 5) << np.uint64(7)) | cubecell(d, 4)
        keys["morton7|cube4"] = (morton(o, 7) << np.uint64(7)) | cubecell(d, 4)
        keys["cube2|morton7"] = (cubecell(d, 2) << np.uint64(21)) | morton(o, 7)
        keys["morton6|cube8"] = (morton(o, 6) << np.uint64(9)) | cubecell(d, 8)
        for kname, key in keys.items():
            order = np.argsort(key, kind="stable")
            v = run(kname, r[:, order], anyhit)
            if kname in ("morton4", "morton8", "morton6|cube8"):
                run_queued(kname, r, order, anyhit)
        run_queued("render order", r, np.arange(n), anyhit)
        # segment-local sort: what a per-chunk (not global) sort would give
        for seg in (1 << 16, 1 << 20):
            key = keys["morton6|oct"]
            order = np.lexsort((key, np.arange(n) // seg))
            run("morton6|oct within %d" % seg, r[:, order], anyhit)
