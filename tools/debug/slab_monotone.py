"""Check, on a real BVH, the property the wide traversal kernels rest on ("Leaf-exact walks"):

    Bounds3::IntersectP(ray, invDir, dirIsNeg) (core/geometry.h:1754-1780) is monotone under box inclusion —
    if a node's box passes, the box of every ancestor (a superset, exactly: interior bounds are the Union of the
    primitive bounds below them, accelerators/bvh.cpp:220-222) passes too, for the same ray and tMax.

So the set of leaves the reference's walk reaches is exactly the set of leaves whose OWN box passes, and any
hierarchy over those leaves that never culls a passing leaf visits the same primitives.  The script evaluates the
slab test (float32, the reference's operation order) on EVERY node of the killeroo fixture's BVH for rays chosen to
hit the degenerate cases: directions with zero components (1/0 = inf, 0 * inf = NaN), origins exactly on node
planes, finite tMax, and reports any node that passes while its parent fails.
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
hprt = importlib.import_module("thesis-pbrt-v3_amd")

f32 = np.float32
GAMMA3 = f32(3) * f32(2 ** -24) / (f32(1) - f32(3) * f32(2 ** -24))
ROBUST = f32(1) + f32(2) * GAMMA3


def slab_pass(lo, hi, o, d, tmax):
    """Vectorised over nodes: lo, hi (n, 3) float32; one ray.  Mirrors the reference statement by statement."""
    with np.errstate(all="ignore"):
        inv = (f32(1) / d).astype(f32)
        neg = inv < 0
        near = np.where(neg[None, :], hi, lo)
        far = np.where(neg[None, :], lo, hi)
        tn = ((near - o[None, :]).astype(f32) * inv[None, :]).astype(f32)
        tf = ((far - o[None, :]).astype(f32) * inv[None, :]).astype(f32)
        tf = (tf * ROBUST).astype(f32)
        tMin, tMax = tn[:, 0].copy(), tf[:, 0].copy()
        ok = ~((tMin > tf[:, 1]) | (tn[:, 1] > tMax))
        m = tn[:, 1] > tMin; tMin[m] = tn[m, 1]
        m = tf[:, 1] < tMax; tMax[m] = tf[m, 1]
        ok &= ~((tMin > tf[:, 2]) | (tn[:, 2] > tMax))
        m = tn[:, 2] > tMin; tMin[m] = tn[m, 2]
        m = tf[:, 2] < tMax; tMax[m] = tf[m, 2]
        return ok & (tMin < tmax) & (tMax > 0)


def main():
    model = hprt.Model.load(os.path.join(ROOT, "tests", "golden", "killeroo_simple.hprt"))
    nodes, _ = hprt.Bvh(model).arrays()
    n = nodes.shape[0]
    lo = nodes[:, 0:3].view(f32).copy(); hi = nodes[:, 3:6].view(f32).copy()
    leaf = (nodes[:, 7] & 3) == 3
    parent = np.full(n, -1, np.int64)
    inter = np.nonzero(~leaf)[0]
    parent[inter + 1] = inter
    parent[nodes[inter, 6]] = inter
    assert (parent[1:] >= 0).all()
    # inclusion is exact
    assert (lo[parent[1:]] <= lo[1:]).all() and (hi[parent[1:]] >= hi[1:]).all()
    rng = np.random.default_rng(7)
    planes = np.concatenate([lo.ravel(), hi.ravel()])
    bad = 0
    nrays = int(os.environ.get("RAYS", "3000"))
    for r in range(nrays):
        kind = r % 6
        o = rng.uniform(lo[0] - 50, hi[0] + 50).astype(f32)
        d = rng.normal(size=3).astype(f32)
        if kind >= 1:      # one or two zero direction components, either sign of zero
            z = rng.choice(3, size=1 + (kind % 2), replace=False)
            d[z] = rng.choice([f32(0.0), f32(-0.0)], size=z.size)
        if kind >= 3:      # origin coordinates exactly on planes of nodes
            k = rng.integers(0, n)
            which = rng.integers(0, 2, 3)
            pick = np.where(which == 1, hi[k], lo[k])
            m = rng.integers(0, 2, 3).astype(bool)
            o[m] = pick[m]
        if kind == 5:
            o[rng.integers(0, 3)] = planes[rng.integers(0, planes.size)]
        tmax = f32(np.inf) if r % 2 == 0 else f32(rng.uniform(0, 400))
        p = slab_pass(lo, hi, o, d, tmax)
        viol = np.nonzero(p[1:] & ~p[parent[1:]])[0]
        if viol.size:
            bad += viol.size
            k = viol[0] + 1
            print("VIOLATION ray", r, "o", o, "d", d, "tmax", tmax, "node", k, lo[k], hi[k], "parent", parent[k], lo[parent[k]], hi[parent[k]])
    print("%d rays x %d nodes: %d nodes passed under a failing parent" % (nrays, n, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
