"""The leaf-exact four-wide walk (csrc/wide_bvh.cpp, k_walk4) on the CPU: what makes it return what the reference's walk returns.

(1) Bounds3::IntersectP (core/geometry.h:1754-1780) is monotone under box inclusion on the real tree — so the reference's walk
    (accelerators/bvh.cpp:354-437) reaches exactly the leaves whose own box passes;
(2) every dequantised child box of every wide record contains the exact box of the node it stands for;
(3) a walk over the wide records that culls with the dequantised boxes, visits slots in the nested near / far order of the
    collapsed split axes and applies the exact box test at the leaves reaches the SAME leaves in the SAME order as the
    reference's walk — for rays with zero direction components, origins on node planes and finite tMax among them.
GPU parity of the kernel itself: tests/test_gpu_parity.py::test_wide_and_binary_walks_agree and every film test (plain renders
take the wide walk, counting renders the binary one, both are held to the oracle)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT

f32 = np.float32
GAMMA3 = f32(3) * f32(2 ** -24) / (f32(1) - f32(3) * f32(2 ** -24))
ROBUST = f32(1) + f32(2) * GAMMA3
NONE = -0x80000000


def slab(lo, hi, o, inv, neg, tmax):
    """The reference's test, statement by statement, in float32.  lo, hi: (..., 3)."""
    with np.errstate(all="ignore"):
        near = np.where(neg, hi, lo); far = np.where(neg, lo, hi)
        tn = ((near - o).astype(f32) * inv).astype(f32)
        tf = (((far - o).astype(f32) * inv).astype(f32) * ROBUST).astype(f32)
        tMin, tMax = tn[..., 0].copy(), tf[..., 0].copy()
        ok = ~((tMin > tf[..., 1]) | (tn[..., 1] > tMax))
        tMin = np.where(tn[..., 1] > tMin, tn[..., 1], tMin); tMax = np.where(tf[..., 1] < tMax, tf[..., 1], tMax)
        ok &= ~((tMin > tf[..., 2]) | (tn[..., 2] > tMax))
        tMin = np.where(tn[..., 2] > tMin, tn[..., 2], tMin); tMax = np.where(tf[..., 2] < tMax, tf[..., 2], tMax)
        return ok & (tMin < tmax) & (tMax > 0)


def conservative(lo, hi, o, inv, neg, tmax):
    """k_walk4's interior test on dequantised boxes: the same operations, rejections as 'provably outside' only."""
    with np.errstate(all="ignore"):
        near = np.where(neg, hi, lo); far = np.where(neg, lo, hi)
        tn = ((near - o).astype(f32) * inv).astype(f32)
        tf = (((far - o).astype(f32) * inv).astype(f32) * ROBUST).astype(f32)
        tE = np.fmax(np.fmax(tn[..., 0], tn[..., 1]), tn[..., 2]); tX = np.fmin(np.fmin(tf[..., 0], tf[..., 1]), tf[..., 2])
        return ~((tE > tX) | (tX <= 0) | (tE >= tmax))


@pytest.fixture(scope="module")
def tree(hprt):
    model = hprt.Model.load(os.path.join(ROOT, "tests", "golden", "killeroo_simple.hprt"))
    nodes, _ = hprt.Bvh(model).arrays()
    fn = hprt.lib.hprt_debug_wide_build
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
    n_out, need = C.c_size_t(0), C.c_int(0)
    wide = np.zeros((nodes.shape[0], 16), np.uint32)
    assert fn(nodes.ctypes.data, nodes.shape[0], wide.ctypes.data, wide.shape[0], C.byref(n_out), C.byref(need)) == 0
    return nodes, wide[:n_out.value].copy(), need.value


def rays(nodes, n, seed):
    lo = nodes[:, 0:3].view(f32); hi = nodes[:, 3:6].view(f32)
    rng = np.random.default_rng(seed)
    out = []
    for r in range(n):
        kind = r % 6
        o = rng.uniform(lo[0] - 30, hi[0] + 30).astype(f32)
        d = rng.normal(size=3).astype(f32)
        if kind in (1, 2, 4):
            z = rng.choice(3, size=1 + (kind % 2), replace=False)
            d[z] = rng.choice([f32(0.0), f32(-0.0)], size=z.size)
        if kind >= 3:
            k = rng.integers(0, nodes.shape[0]); m = rng.integers(0, 2, 3).astype(bool)
            o[m] = np.where(rng.integers(0, 2, 3) == 1, hi[k], lo[k])[m]
        if kind == 0:      # aimed at a leaf, so that deep parts of the tree are walked
            k = rng.integers(0, nodes.shape[0])
            d = ((lo[k] + hi[k]) * f32(0.5) - o).astype(f32)
        tmax = f32(np.inf) if r % 2 == 0 else f32(rng.uniform(0, 300))
        out.append((o, d, tmax))
    return out


def test_the_slab_test_is_monotone_under_inclusion(tree):
    nodes, _, _ = tree
    lo = nodes[:, 0:3].view(f32); hi = nodes[:, 3:6].view(f32)
    leaf = (nodes[:, 7] & 3) == 3
    inter = np.nonzero(~leaf)[0]
    parent = np.full(nodes.shape[0], -1, np.int64)
    parent[inter + 1] = inter; parent[nodes[inter, 6]] = inter
    assert (lo[parent[1:]] <= lo[1:]).all() and (hi[parent[1:]] >= hi[1:]).all()      # interior bounds are exact unions
    for o, d, tmax in rays(nodes, 300, 3):
        with np.errstate(all="ignore"):
            inv = (f32(1) / d).astype(f32)
        p = slab(lo, hi, o, inv, inv < 0, tmax)
        assert not (p[1:] & ~p[parent[1:]]).any()


def dequant(wide):
    """(n, 4, 3) lo and hi of every slot, float32, as the kernel forms them: origin + q * 2^e (q * 2^e is exact)."""
    org = wide[:, 0:3].view(f32)
    em = wide[:, 3]
    step = np.stack([((em >> (8 * a)) & 0xff).astype(np.uint32) << 23 for a in range(3)], axis=1).view(f32)
    lo = np.zeros((wide.shape[0], 4, 3), f32); hi = np.zeros_like(lo)
    for a in range(3):
        for s in range(4):
            ql = ((wide[:, 4 + 2 * a] >> (8 * s)) & 0xff).astype(f32); qh = ((wide[:, 5 + 2 * a] >> (8 * s)) & 0xff).astype(f32)
            lo[:, s, a] = (org[:, a] + (ql * step[:, a]).astype(f32)).astype(f32)
            hi[:, s, a] = (org[:, a] + (qh * step[:, a]).astype(f32)).astype(f32)
    return lo, hi


def slot_nodes(nodes, wide):
    """Binary node every slot of every wide record stands for (-1: empty), by replaying the collapse."""
    leaf = (nodes[:, 7] & 3) == 3
    ref = wide[:, 12:16].view(np.int32)
    of = np.full((wide.shape[0], 4), -1, np.int64)
    if leaf[0]:      # a one-leaf tree: one record whose only slot is that leaf
        of[0, 0] = 0
        return of
    todo = [(0, 0)]
    while todo:
        n, w = todo.pop()
        c = (n + 1, int(nodes[n, 6]))
        for g in range(2):
            k = c[g]
            kids = [k] if leaf[k] else [k + 1, int(nodes[k, 6])]
            for j, kid in enumerate(kids):
                of[w, 2 * g + j] = kid
                if not leaf[kid]:
                    todo.append((kid, int(ref[w, 2 * g + j])))
    return of


def test_dequantised_boxes_contain_the_exact_ones(tree):
    nodes, wide, need = tree
    assert 0 < need <= 60
    lo = nodes[:, 0:3].view(f32); hi = nodes[:, 3:6].view(f32)
    of = slot_nodes(nodes, wide)
    ref = wide[:, 12:16].view(np.int32)
    assert ((of >= 0) == (ref != NONE)).all()
    qlo, qhi = dequant(wide)
    m = of >= 0
    assert (qlo[m] <= lo[of[m]]).all() and (qhi[m] >= hi[of[m]]).all()
    # and not by much: within two grid steps
    step = np.stack([((wide[:, 3] >> (8 * a)) & 0xff).astype(np.uint32) << 23 for a in range(3)], axis=1).view(f32)
    st = np.broadcast_to(step[:, None, :], qlo.shape)
    assert (lo[of[m]] - qlo[m] <= 2 * st[m]).all() and (qhi[m] - hi[of[m]] <= 2 * st[m]).all()
    # every leaf of the tree is some slot, exactly once
    leaf_ids = np.nonzero((nodes[:, 7] & 3) == 3)[0]
    seen = np.sort(of[m & (ref < 0)])
    assert np.array_equal(seen, leaf_ids)


def test_the_wide_walk_reaches_the_same_leaves_in_the_same_order(tree):
    nodes, wide, _ = tree
    lo = nodes[:, 0:3].view(f32); hi = nodes[:, 3:6].view(f32)
    leaf = (nodes[:, 7] & 3) == 3
    axis = nodes[:, 7] & 3
    qlo, qhi = dequant(wide)
    of = slot_nodes(nodes, wide)
    ref = wide[:, 12:16].view(np.int32)
    meta = wide[:, 3] >> 24
    longest = 0
    for o, d, tmax in rays(nodes, 240, 5):
        with np.errstate(all="ignore"):
            inv = (f32(1) / d).astype(f32)
        neg = inv < 0
        # the reference's walk (bvh.cpp:354-437) with a tMax that does not shrink
        want, stack, cur = [], [], 0
        while True:
            if slab(lo[cur], hi[cur], o, inv, neg, tmax):
                if leaf[cur]:
                    want.append(cur)
                    if not stack: break
                    cur = stack.pop()
                elif neg[axis[cur]]:
                    stack.append(cur + 1); cur = int(nodes[cur, 6])
                else:
                    stack.append(int(nodes[cur, 6])); cur = cur + 1
            else:
                if not stack: break
                cur = stack.pop()
        # the wide walk
        got, stack, cur = [], [], 0
        while True:
            if cur >= 0:
                ok = conservative(qlo[cur], qhi[cur], o, inv, neg, tmax) & (ref[cur] != NONE)
                m = int(meta[cur])
                order = [0, 1, 2, 3]
                if neg[(m >> 2) & 3]: order[0], order[1] = order[1], order[0]
                if neg[(m >> 4) & 3]: order[2], order[3] = order[3], order[2]
                if neg[m & 3]: order = order[2:] + order[:2]
                hit = [s for s in order if ok[s]]
                for s in reversed(hit[1:]):
                    stack.append((int(ref[cur, s]), int(of[cur, s])))
                nxt = (int(ref[cur, hit[0]]), int(of[cur, hit[0]])) if hit else (stack.pop() if stack else None)
            else:
                if slab(lo[node], hi[node], o, inv, neg, tmax): got.append(node)      # the leaf's own exact box
                nxt = stack.pop() if stack else None
            if nxt is None: break
            cur, node = nxt
        assert got == want, (o, d, tmax)
        longest = max(longest, len(want))
    assert longest >= 3


def test_a_tree_that_does_not_fit_the_grid_keeps_the_binary_walk(hprt, tree):
    """BuildWide refuses what it cannot bound conservatively; hprt_scene_create then uploads no wide records and every call takes k_trace."""
    nodes, _, _ = tree
    fn = hprt.lib.hprt_debug_wide_build
    n_out, need = C.c_size_t(0), C.c_int(0)
    leaf = int(np.nonzero((nodes[:, 7] & 3) == 3)[0][5])
    for value in (np.inf, np.nan):
        bad = nodes.copy()
        bad[leaf, 4] = np.array([value], np.float32).view(np.uint32)[0]      # one leaf's upper y bound
        assert fn(bad.ctypes.data, bad.shape[0], None, 0, C.byref(n_out), C.byref(need)) == hprt.E_UNSUPPORTED
    # an inverted box (lower bound above the upper one: nothing the builder produces) is refused too; any finite extent fits the grid
    # (255 steps of 2^127 cover the float range)
    inv = nodes.copy()
    inv[leaf, 1], inv[leaf, 4] = nodes[leaf, 4], nodes[leaf, 1]
    if nodes[leaf, 1] != nodes[leaf, 4]:
        assert fn(inv.ctypes.data, inv.shape[0], None, 0, C.byref(n_out), C.byref(need)) == hprt.E_UNSUPPORTED
    wide_span = nodes.copy()
    lo = wide_span[:, 0:3].view(f32); hi = wide_span[:, 3:6].view(f32)
    lo[1, 0] = f32(-3e38); hi[int(wide_span[0, 6]), 0] = f32(3e38)
    assert fn(wide_span.ctypes.data, wide_span.shape[0], None, 0, C.byref(n_out), C.byref(need)) == 0


@pytest.mark.parametrize("seed,scale,offset", [(1, 1.0, 0.0), (2, 1e-30, 0.0), (3, 1e25, 0.0), (4, 1e-3, 1e6), (5, 1.0, -3e30)])
def test_enclosure_holds_on_random_trees_at_extreme_scales(hprt, seed, scale, offset):
    """BuildWide on trees of random boxes — flat and point-sized boxes among them, coordinates from the denormal range to 1e30, small
    extents far from the origin (where one grid step is below the float spacing): every dequantised slot still contains its node's box."""
    rng = np.random.default_rng(seed)
    n = 3000
    c = rng.uniform(-1, 1, (n, 3)); h = rng.uniform(0, 0.05, (n, 3))
    h[rng.integers(0, n, n // 5), rng.integers(0, 3, n // 5)] = 0.0      # flat boxes
    h[rng.integers(0, n, n // 20)] = 0.0                                 # points
    lo_d = (c - h) * scale + offset; hi_d = (c + h) * scale + offset
    bmin = lo_d.astype(f32); bmax = np.maximum(hi_d.astype(f32), bmin)
    nodes, _ = hprt.Bvh.from_bounds(bmin, bmax).arrays()
    fn = hprt.lib.hprt_debug_wide_build
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
    n_out, need = C.c_size_t(0), C.c_int(0)
    wide = np.zeros((nodes.shape[0], 16), np.uint32)
    assert fn(nodes.ctypes.data, nodes.shape[0], wide.ctypes.data, wide.shape[0], C.byref(n_out), C.byref(need)) == 0
    wide = wide[:n_out.value]
    lo = nodes[:, 0:3].view(f32); hi = nodes[:, 3:6].view(f32)
    of = slot_nodes(nodes, wide)
    qlo, qhi = dequant(wide)
    m = of >= 0
    assert np.isfinite(qlo[m]).all() and np.isfinite(qhi[m]).all()
    assert (qlo[m] <= lo[of[m]]).all() and (qhi[m] >= hi[of[m]]).all()
    assert np.array_equal(np.sort(of[m & (wide[:, 12:16].view(np.int32) < 0)]), np.nonzero((nodes[:, 7] & 3) == 3)[0])
