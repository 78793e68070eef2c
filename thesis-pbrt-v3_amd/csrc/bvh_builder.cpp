// hprt host side — sweep-SAH BVH builder (see bvh_builder.h for the contract).
//
// Same decisions as the reference's iterativeBuild (accelerators/bvh.cpp:196-333)
// but organised for large inputs: instead of re-sorting the centroids of every node
// on every axis (O(n log^2 n)) the primitives are sorted ONCE per axis by the
// reference's total order (centroid, primitive number) and each split stably
// partitions the three sorted ranges, so every node sees exactly the sequence
// std::sort would have produced.  A fourth array tracks the reference's
// `primitiveInfo` order, which is permuted by libstdc++'s bidirectional
// std::partition (the order primitives get inside multi-primitive leaves).
#include <algorithm>
#include <cmath>
#include <limits>
#include "bvh_builder.h"
#include "host_transform.h"

namespace hprt {
namespace {

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int k = 0; k < 3; ++k) { lo[k] = std::numeric_limits<float>::max(); hi[k] = std::numeric_limits<float>::lowest(); }
    }
    void grow(const float *blo, const float *bhi) {   // Union(Bounds3, Bounds3): std::min / std::max per component
        for (int k = 0; k < 3; ++k) { lo[k] = sel_min(lo[k], blo[k]); hi[k] = sel_max(hi[k], bhi[k]); }
    }
    float area() const {                               // Bounds3::SurfaceArea, geometry.h:946-949
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return 2 * (dx * dy + dx * dz + dy * dz);
    }
};

struct TempNode { Box box; int child[2]; uint32_t axis, first, count; bool leaf; };
struct Work { int node, start, end; };

}  // namespace

// Transform::operator()(const Bounds3f &), core/transform.cpp:238-249: corners in that order
static void TransformBounds(const mat4 &M, vec3 lo, vec3 hi, float *outLo, float *outHi) {
    const vec3 corner[8] = {vec3(lo.x, lo.y, lo.z), vec3(hi.x, lo.y, lo.z), vec3(lo.x, hi.y, lo.z), vec3(lo.x, lo.y, hi.z),
                            vec3(lo.x, hi.y, hi.z), vec3(hi.x, hi.y, lo.z), vec3(hi.x, lo.y, hi.z), vec3(hi.x, hi.y, hi.z)};
    vec3 p0 = xf_point(M, corner[0]);
    vec3 mn = p0, mx = p0;
    for (int c = 1; c < 8; ++c) { vec3 p = xf_point(M, corner[c]); mn = vmin(mn, p); mx = vmax(mx, p); }
    outLo[0] = mn.x; outLo[1] = mn.y; outLo[2] = mn.z; outHi[0] = mx.x; outHi[1] = mx.y; outHi[2] = mx.z;
}
static void ShapePrimBounds(const ShapeDesc &sh, std::vector<float> *bmin, std::vector<float> *bmax) {
    const size_t base = bmin->size() / 3;
    const size_t n = sh.nPrims();
    bmin->resize(3 * (base + n)); bmax->resize(3 * (base + n));
    float *lo = bmin->data() + 3 * base, *hi = bmax->data() + 3 * base;
    if (sh.kind == kTriangleMesh) {
        const MeshData &m = sh.mesh;
        for (uint32_t t = 0; t < m.nTris(); ++t, lo += 3, hi += 3) {
            const float *a = &m.P[3 * (size_t)m.indices[3 * t]], *b = &m.P[3 * (size_t)m.indices[3 * t + 1]],
                        *c = &m.P[3 * (size_t)m.indices[3 * t + 2]];
            for (int d = 0; d < 3; ++d) {
                float l = sel_min(a[d], b[d]), h = sel_max(a[d], b[d]);   // Bounds3f(p0, p1)
                lo[d] = sel_min(l, c[d]);                                 // Union(b, p2)
                hi[d] = sel_max(h, c[d]);
            }
        }
    } else {
        // Transform::operator()(Bounds3f) on Sphere::ObjectBound (shapes/sphere.cpp:43-46)
        const SphereData &s = sh.sphere;
        TransformBounds(s.objectToWorld, vec3(-s.radius, -s.radius, s.zMin), vec3(s.radius, s.radius, s.zMax), lo, hi);
    }
}
void ComputeObjectPrimBounds(const SceneModel &sc, int object, std::vector<float> *bmin, std::vector<float> *bmax) {
    bmin->clear(); bmax->clear();
    for (const ShapeDesc &sh : sc.shapes) if (sh.object == object) ShapePrimBounds(sh, bmin, bmax);
}
void ComputePrimBounds(const SceneModel &sc, const std::vector<BvhTree> &objectTrees, std::vector<float> *bmin,
                       std::vector<float> *bmax) {
    bmin->clear(); bmax->clear();
    for (const TopItem &t : sc.top) {
        if (t.kind == 0) { ShapePrimBounds(sc.shapes[t.index], bmin, bmax); continue; }
        // TransformedPrimitive::WorldBound = PrimitiveToWorld.MotionBounds(primitive->WorldBound()) (core/primitive.h:116-118);
        // without animation: the instance transform applied to the bounds of the object's aggregate (or lone primitive)
        const InstanceDesc &in = sc.instances[t.index];
        const BvhTree &tree = objectTrees[(size_t)in.object];
        float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        if (!tree.nodes.empty())
            TransformBounds(in.instanceToWorld, vec3(tree.nodes[0].bmin[0], tree.nodes[0].bmin[1], tree.nodes[0].bmin[2]),
                            vec3(tree.nodes[0].bmax[0], tree.nodes[0].bmax[1], tree.nodes[0].bmax[2]), lo, hi);
        bmin->insert(bmin->end(), lo, lo + 3); bmax->insert(bmax->end(), hi, hi + 3);
    }
}

void BuildBvh(size_t nPrims, const float *bmin, const float *bmax, int maxPrimsInNodeIn, int isectCost, int travCost,
              BvhTree *out) {
    out->nodes.clear(); out->primOrder.clear(); out->maxDepth = 0; out->nLeaves = 0;
    if (nPrims == 0) return;
    const int maxPrimsInNode = std::min(255, maxPrimsInNodeIn);   // bvh.cpp:157
    const uint32_t n = (uint32_t)nPrims;
    // centroid = .5f*pMin + .5f*pMax (bvh.cpp:59), kept per axis
    std::vector<float> cen[3];
    for (int a = 0; a < 3; ++a) {
        cen[a].resize(n);
        for (uint32_t i = 0; i < n; ++i) cen[a][i] = .5f * bmin[3 * (size_t)i + a] + .5f * bmax[3 * (size_t)i + a];
    }
    // one global sort per axis by (centroid, prim number) — the comparator of bvh.cpp:251-257
    std::vector<uint32_t> sorted[3];
    for (int a = 0; a < 3; ++a) {
        sorted[a].resize(n);
        for (uint32_t i = 0; i < n; ++i) sorted[a][i] = i;
        const float *c = cen[a].data();
        std::sort(sorted[a].begin(), sorted[a].end(), [c](uint32_t x, uint32_t y) {
            if (c[x] == c[y]) return (int)x < (int)y;
            return c[x] < c[y];
        });
    }
    std::vector<uint32_t> info(n);             // the reference's primitiveInfo order
    for (uint32_t i = 0; i < n; ++i) info[i] = i;
    std::vector<uint32_t> scratch(n);
    std::vector<float> rightArea(n + 1);
    std::vector<uint8_t> goesLeft(n);

    std::vector<TempNode> tmp;
    tmp.reserve(2 * (size_t)n);
    tmp.push_back(TempNode());
    std::vector<Work> stack;
    stack.push_back(Work{0, 0, (int)n});
    while (!stack.empty()) {
        Work w = stack.back();
        stack.pop_back();
        const uint32_t count = (uint32_t)(w.end - w.start);
        Box bounds; bounds.reset();
        for (int i = w.start; i < w.end; ++i) bounds.grow(&bmin[3 * (size_t)info[i]], &bmax[3 * (size_t)info[i]]);
        if (count == 1) {
            TempNode &nd = tmp[w.node];
            nd.leaf = true; nd.first = (uint32_t)out->primOrder.size(); nd.count = 1; nd.box = bounds; nd.child[0] = nd.child[1] = -1;
            out->primOrder.push_back(info[w.start]);
            continue;
        }
        int bestAxis = -1; uint32_t bestPrim = 0; Box bestBox; bestBox.reset();
        float bestCost = HPRT_INF;
        const float oldCost = isectCost * float(count);
        const float totalSA = bounds.area();
        const float invTotalSA = 1 / totalSA;
        for (int a = 0; a < 3; ++a) {
            const uint32_t *ord = &sorted[a][w.start];
            Box acc; acc.reset();
            for (int i = (int)count - 1; i >= 0; --i) {   // suffix bounds (rightToLeftBounds)
                acc.grow(&bmin[3 * (size_t)ord[i]], &bmax[3 * (size_t)ord[i]]);
                rightArea[i] = acc.area();
            }
            const Box whole = acc;
            Box left; left.reset();
            for (int i = 0; i < (int)count - 1; ++i) {   // prefix bounds + cost, bvh.cpp:266-288
                left.grow(&bmin[3 * (size_t)ord[i]], &bmax[3 * (size_t)ord[i]]);
                float cost = travCost + isectCost * ((i + 1) * left.area() + (count - i - 1) * rightArea[i + 1]) * invTotalSA;
                if (cost < bestCost) { bestCost = cost; bestAxis = a; bestPrim = ord[i]; bestBox = whole; }
            }
        }
        if (bestAxis != -1 && (bestCost < oldCost || count > (uint32_t)maxPrimsInNode)) {
            const float bc = cen[bestAxis][bestPrim];
            const float *c = cen[bestAxis].data();
            auto pred = [&](uint32_t p) { return c[p] < bc || (c[p] == bc && p <= bestPrim); };
            // libstdc++ std::__partition for bidirectional iterators, applied to `info`
            uint32_t *first = &info[w.start], *last = &info[w.start] + count;
            while (true) {
                while (first != last && pred(*first)) ++first;
                if (first == last) break;
                --last;
                while (first != last && !pred(*last)) --last;
                if (first == last) break;
                std::swap(*first, *last);
                ++first;
            }
            const int mid = (int)(first - &info[0]);
            // stable partition of the three sorted ranges
            for (int i = w.start; i < w.end; ++i) goesLeft[info[i]] = (i < mid) ? 1 : 0;
            for (int a = 0; a < 3; ++a) {
                uint32_t *ord = &sorted[a][w.start];
                uint32_t nl = 0, nr = 0;
                for (uint32_t i = 0; i < count; ++i) {
                    if (goesLeft[ord[i]]) ord[nl++] = ord[i];
                    else scratch[nr++] = ord[i];
                }
                std::copy(scratch.begin(), scratch.begin() + nr, ord + nl);
            }
            int c0 = (int)tmp.size(); tmp.push_back(TempNode());
            int c1 = (int)tmp.size(); tmp.push_back(TempNode());
            TempNode &nd = tmp[w.node];
            nd.leaf = false; nd.axis = (uint32_t)bestAxis; nd.count = count; nd.box = bestBox; nd.child[0] = c0; nd.child[1] = c1; nd.first = 0;
            stack.push_back(Work{c0, w.start, mid});     // c1 is popped (built) first, bvh.cpp:308-309
            stack.push_back(Work{c1, mid, w.end});
        } else {
            TempNode &nd = tmp[w.node];
            nd.leaf = true; nd.first = (uint32_t)out->primOrder.size(); nd.count = count; nd.box = bounds; nd.child[0] = nd.child[1] = -1;
            for (int i = w.start; i < w.end; ++i) out->primOrder.push_back(info[i]);
        }
    }
    // depth-first flattening, first child adjacent (bvh.cpp:335-350); iterative
    out->nodes.resize(tmp.size());
    struct Visit { int tnode; int parentSlot; int depth; };
    std::vector<Visit> vs;
    vs.push_back(Visit{0, -1, 1});
    int next = 0;
    while (!vs.empty()) {
        Visit v = vs.back(); vs.pop_back();
        const TempNode &t = tmp[v.tnode];
        int slot = next++;
        if (v.parentSlot >= 0) out->nodes[v.parentSlot].offset = slot;   // this is a second child
        if (v.depth > out->maxDepth) out->maxDepth = v.depth;
        BvhNode &ln = out->nodes[slot];
        for (int k = 0; k < 3; ++k) { ln.bmin[k] = t.box.lo[k]; ln.bmax[k] = t.box.hi[k]; }
        if (t.leaf) {
            ++out->nLeaves;
            ln.countAxis = 3u | (t.count << 2);
            ln.offset = (int32_t)t.first;
        } else {
            ln.countAxis = t.axis | (t.count << 2);
            ln.offset = 0;
            vs.push_back(Visit{t.child[1], slot, v.depth + 1});   // visited after the whole child[0] subtree
            vs.push_back(Visit{t.child[0], -1, v.depth + 1});
        }
    }
}

}  // namespace hprt
