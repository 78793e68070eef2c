// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
// Shapes and the BVH: shapes/triangle.cpp, shapes/sphere.cpp, core/efloat.h,
// accelerators/bvh.cpp, core/primitive.cpp, core/interaction.{h,cpp}.
#pragma once
#include <algorithm>
#include <atomic>
#include <vector>
#include "orc_scene.h"

namespace orc {

// Per-thread traversal counters (the reference's STAT_COUNTERs:
// accelerators/bvh.cpp:48-49, shapes/triangle.cpp:43-44, core/scene.cpp:40-42)
// plus the *fetched* node count V that SURVEY.md §8(d) defines.
struct Counters {
    uint64_t nodesFetched = 0, nodesFetchedP = 0;    // traversal-loop iterations
    uint64_t nodesEntered = 0, nodesEnteredP = 0;    // slab test passed (reference counter)
    uint64_t leavesEntered = 0, leavesEnteredP = 0;  // of those, leaves: ray.stats.leafNodeTraversals[P] (bvh.cpp:370,411)
    uint64_t triTests = 0, triTestsP = 0, triHits = 0, triHitsP = 0;
    uint64_t sphereTests = 0, sphereTestsP = 0;
    // Shape::Pdf's own Intersect call on an emitter (core/shape.cpp:72-88) does not go through GeometricPrimitive: the fork's
    // per-ray primitiveIntersections (core/primitive.cpp:119-125) do not see it, the reference's global nTests of
    // Triangle::Intersect (shapes/triangle.cpp:191) does.  Kept apart so that both statistics can be formed.
    uint64_t triTestsPdf = 0, triHitsPdf = 0, sphereTestsPdf = 0;
    uint64_t rays = 0, shadowRays = 0, cameraRays = 0;
    void add(const Counters &o) {
        nodesFetched += o.nodesFetched; nodesFetchedP += o.nodesFetchedP;
        nodesEntered += o.nodesEntered; nodesEnteredP += o.nodesEnteredP;
        leavesEntered += o.leavesEntered; leavesEnteredP += o.leavesEnteredP;
        triTests += o.triTests; triTestsP += o.triTestsP; triHits += o.triHits; triHitsP += o.triHitsP;
        sphereTests += o.sphereTests; sphereTestsP += o.sphereTestsP;
        triTestsPdf += o.triTestsPdf; triHitsPdf += o.triHitsPdf; sphereTestsPdf += o.sphereTestsPdf;
        rays += o.rays; shadowRays += o.shadowRays; cameraRays += o.cameraRays;
    }
};

// core/interaction.h:51-157 — the members the path integrator reads.
struct SurfaceInteraction {
    V3 p, pError, wo, n;
    P2 uv;
    V3 dpdu, dpdv;
    struct { V3 n, dpdu, dpdv; } shading;
    int prim = -1;        // creation-order primitive number (within its aggregate)
    int shape = -1;
    int tri = -1;         // triangle number within its mesh (-1: not a triangle): which DiffuseAreaLight an emissive mesh hit belongs to
    Float dudx = 0, dvdx = 0, dudy = 0, dvdy = 0;      // ComputeDifferentials (core/interaction.cpp:103-149)
    int ordered = -1;     // ordered index of the hit primitive over all aggregates: top level first, then object 0, 1, ...
    int inst = -1;        // instance the hit went through (Scene::instances), -1: none
    // barycentrics (triangles) kept for the per-ray golden vectors
    Float b0 = 0, b1 = 0, b2 = 0;
};

// SurfaceInteraction ctor, core/interaction.cpp:43-70
inline void InitSI(SurfaceInteraction *si, const V3 &p, const V3 &pError, const P2 &uv, const V3 &wo,
                   const V3 &dpdu, const V3 &dpdv, bool flip) {
    si->p = p; si->pError = pError; si->uv = uv;
    si->wo = Normalize(wo);                       // interaction.h:61
    si->n = Normalize(Cross(dpdu, dpdv));
    si->dpdu = dpdu; si->dpdv = dpdv;
    si->shading.n = si->n; si->shading.dpdu = dpdu; si->shading.dpdv = dpdv;
    if (flip) { si->n *= -1; si->shading.n *= -1; }
}
// core/interaction.cpp:72-91
inline void SetShadingGeometry(SurfaceInteraction *si, const V3 &dpdus, const V3 &dpdvs, bool flip,
                               bool orientationIsAuthoritative) {
    si->shading.n = Normalize(Cross(dpdus, dpdvs));
    if (flip) si->shading.n = -si->shading.n;
    if (orientationIsAuthoritative) si->n = Faceforward(si->n, si->shading.n);
    else si->shading.n = Faceforward(si->shading.n, si->n);
    si->shading.dpdu = dpdus; si->shading.dpdv = dpdvs;
}

struct TriRef { const Mesh *mesh; const int *v; bool flip; };

inline void GetUVs(const TriRef &t, P2 uv[3]) {   // shapes/triangle.h:109-119
    if (t.mesh->hasUV) { uv[0] = t.mesh->uv[t.v[0]]; uv[1] = t.mesh->uv[t.v[1]]; uv[2] = t.mesh->uv[t.v[2]]; }
    else { uv[0] = P2(0, 0); uv[1] = P2(1, 0); uv[2] = P2(1, 1); }
}

// Shared head of Triangle::Intersect / IntersectP
// (shapes/triangle.cpp:193-292 == :431-530).  Returns false on a miss.
inline bool TriangleTest(const V3 &p0, const V3 &p1, const V3 &p2, const Ray &ray, Float *b0o, Float *b1o,
                         Float *b2o, Float *to) {
    V3 p0t = p0 - ray.o, p1t = p1 - ray.o, p2t = p2 - ray.o;
    int kz = MaxDimension(Abs(ray.d));
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    V3 d = Permute(ray.d, kx, ky, kz);
    p0t = Permute(p0t, kx, ky, kz); p1t = Permute(p1t, kx, ky, kz); p2t = Permute(p2t, kx, ky, kz);
    Float Sx = -d.x / d.z, Sy = -d.y / d.z, Sz = 1.f / d.z;
    p0t.x += Sx * p0t.z; p0t.y += Sy * p0t.z;
    p1t.x += Sx * p1t.z; p1t.y += Sy * p1t.z;
    p2t.x += Sx * p2t.z; p2t.y += Sy * p2t.z;
    Float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    Float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    Float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
        double p2txp1ty = (double)p2t.x * (double)p1t.y;
        double p2typ1tx = (double)p2t.y * (double)p1t.x;
        e0 = (float)(p2typ1tx - p2txp1ty);
        double p0txp2ty = (double)p0t.x * (double)p2t.y;
        double p0typ2tx = (double)p0t.y * (double)p2t.x;
        e1 = (float)(p0typ2tx - p0txp2ty);
        double p1txp0ty = (double)p1t.x * (double)p0t.y;
        double p1typ0tx = (double)p1t.y * (double)p0t.x;
        e2 = (float)(p1typ0tx - p1txp0ty);
    }
    if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return false;
    Float det = e0 + e1 + e2;
    if (det == 0) return false;
    p0t.z *= Sz; p1t.z *= Sz; p2t.z *= Sz;
    Float tScaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if (det < 0 && (tScaled >= 0 || tScaled < ray.tMax * det)) return false;
    else if (det > 0 && (tScaled <= 0 || tScaled > ray.tMax * det)) return false;
    Float invDet = 1 / det;
    Float b0 = e0 * invDet, b1 = e1 * invDet, b2 = e2 * invDet;
    Float t = tScaled * invDet;
    Float maxZt = MaxComponent(Abs(V3(p0t.z, p1t.z, p2t.z)));
    Float deltaZ = gamma(3) * maxZt;
    Float maxXt = MaxComponent(Abs(V3(p0t.x, p1t.x, p2t.x)));
    Float maxYt = MaxComponent(Abs(V3(p0t.y, p1t.y, p2t.y)));
    Float deltaX = gamma(5) * (maxXt + maxZt);
    Float deltaY = gamma(5) * (maxYt + maxZt);
    Float deltaE = 2 * (gamma(2) * maxXt * maxYt + deltaY * maxXt + deltaX * maxYt);
    Float maxE = MaxComponent(Abs(V3(e0, e1, e2)));
    Float deltaT = 3 * (gamma(3) * maxE * maxZt + deltaE * maxZt + deltaZ * maxE) * std::abs(invDet);
    if (t <= deltaT) return false;
    *b0o = b0; *b1o = b1; *b2o = b2; *to = t;
    return true;
}

// shapes/triangle.cpp:188-426
inline bool TriangleIntersect(const TriRef &tr, const Ray &ray, Float *tHit, SurfaceInteraction *isect,
                              Counters &ctr) {
    ++ctr.triTests;
    const Mesh *mesh = tr.mesh;
    const V3 &p0 = mesh->p[tr.v[0]], &p1 = mesh->p[tr.v[1]], &p2 = mesh->p[tr.v[2]];
    Float b0, b1, b2, t;
    if (!TriangleTest(p0, p1, p2, ray, &b0, &b1, &b2, &t)) return false;
    // partial derivatives (:294-318)
    V3 dpdu, dpdv;
    P2 uv[3];
    GetUVs(tr, uv);
    Float duv02x = uv[0].x - uv[2].x, duv02y = uv[0].y - uv[2].y;
    Float duv12x = uv[1].x - uv[2].x, duv12y = uv[1].y - uv[2].y;
    V3 dp02 = p0 - p2, dp12 = p1 - p2;
    Float determinant = duv02x * duv12y - duv02y * duv12x;
    bool degenerateUV = std::abs(determinant) < 1e-8;
    if (!degenerateUV) {
        Float invdet = 1 / determinant;
        dpdu = (duv12y * dp02 - duv02y * dp12) * invdet;
        dpdv = (-duv12x * dp02 + duv02x * dp12) * invdet;
    }
    if (degenerateUV || Cross(dpdu, dpdv).LengthSquared() == 0) {
        V3 ng = Cross(p2 - p0, p1 - p0);
        if (ng.LengthSquared() == 0) return false;
        CoordinateSystem(Normalize(ng), &dpdu, &dpdv);
    }
    // error bounds (:320-327)
    Float xAbsSum = (std::abs(b0 * p0.x) + std::abs(b1 * p1.x) + std::abs(b2 * p2.x));
    Float yAbsSum = (std::abs(b0 * p0.y) + std::abs(b1 * p1.y) + std::abs(b2 * p2.y));
    Float zAbsSum = (std::abs(b0 * p0.z) + std::abs(b1 * p1.z) + std::abs(b2 * p2.z));
    V3 pError = gamma(7) * V3(xAbsSum, yAbsSum, zAbsSum);
    V3 pHit = b0 * p0 + b1 * p1 + b2 * p2;
    P2 uvHit(b0 * uv[0].x + b1 * uv[1].x + b2 * uv[2].x, b0 * uv[0].y + b1 * uv[1].y + b2 * uv[2].y);
    InitSI(isect, pHit, pError, uvHit, -ray.d, dpdu, dpdv, tr.flip);
    isect->b0 = b0; isect->b1 = b1; isect->b2 = b2;
    // :346-347
    isect->n = isect->shading.n = Normalize(Cross(dp02, dp12));
    if (mesh->hasN || mesh->hasS) {
        V3 ns;
        if (mesh->hasN) {
            ns = (b0 * mesh->n[tr.v[0]] + b1 * mesh->n[tr.v[1]] + b2 * mesh->n[tr.v[2]]);
            if (ns.LengthSquared() > 0) ns = Normalize(ns);
            else ns = isect->n;
        } else ns = isect->n;
        V3 ss;
        if (mesh->hasS) {
            ss = (b0 * mesh->s[tr.v[0]] + b1 * mesh->s[tr.v[1]] + b2 * mesh->s[tr.v[2]]);
            if (ss.LengthSquared() > 0) ss = Normalize(ss);
            else ss = Normalize(isect->dpdu);
        } else ss = Normalize(isect->dpdu);
        V3 ts = Cross(ss, ns);
        if (ts.LengthSquared() > 0.f) { ts = Normalize(ts); ss = Cross(ts, ns); }
        else CoordinateSystem(ns, &ss, &ts);
        SetShadingGeometry(isect, ss, ts, tr.flip, true);
    }
    if (mesh->hasN) isect->n = Faceforward(isect->n, isect->shading.n);
    else if (tr.flip) isect->n = isect->shading.n = -isect->n;
    *tHit = t;
    ++ctr.triHits;
    return true;
}
// shapes/triangle.cpp:428-574 (no alpha masks)
inline bool TriangleIntersectP(const TriRef &tr, const Ray &ray, Counters &ctr) {
    ++ctr.triTestsP;
    const Mesh *mesh = tr.mesh;
    Float b0, b1, b2, t;
    if (!TriangleTest(mesh->p[tr.v[0]], mesh->p[tr.v[1]], mesh->p[tr.v[2]], ray, &b0, &b1, &b2, &t)) return false;
    ++ctr.triHitsP;
    return true;
}

// ---- EFloat (core/efloat.h, NDEBUG build: v, low, high only) -------------
struct EFloat {
    float v, low, high;
    EFloat() {}
    EFloat(float v, float err = 0.f) : v(v) {
        if (err == 0.) low = high = v;
        else { low = NextFloatDown(v - err); high = NextFloatUp(v + err); }
    }
    EFloat operator+(EFloat ef) const {
        EFloat r; r.v = v + ef.v;
        r.low = NextFloatDown(low + ef.low); r.high = NextFloatUp(high + ef.high); return r;
    }
    EFloat operator-(EFloat ef) const {
        EFloat r; r.v = v - ef.v;
        r.low = NextFloatDown(low - ef.high); r.high = NextFloatUp(high - ef.low); return r;
    }
    EFloat operator*(EFloat ef) const {
        EFloat r; r.v = v * ef.v;
        Float prod[4] = {low * ef.low, high * ef.low, low * ef.high, high * ef.high};
        r.low = NextFloatDown(smin(smin(prod[0], prod[1]), smin(prod[2], prod[3])));
        r.high = NextFloatUp(smax(smax(prod[0], prod[1]), smax(prod[2], prod[3])));
        return r;
    }
    EFloat operator/(EFloat ef) const {
        EFloat r; r.v = v / ef.v;
        if (ef.low < 0 && ef.high > 0) { r.low = -Infinity; r.high = Infinity; }
        else {
            Float div[4] = {low / ef.low, high / ef.low, low / ef.high, high / ef.high};
            r.low = NextFloatDown(smin(smin(div[0], div[1]), smin(div[2], div[3])));
            r.high = NextFloatUp(smax(smax(div[0], div[1]), smax(div[2], div[3])));
        }
        return r;
    }
    bool operator==(EFloat fe) const { return v == fe.v; }
};
inline EFloat operator*(float f, EFloat fe) { return EFloat(f) * fe; }
// core/efloat.h:267-288
inline bool Quadratic(EFloat A, EFloat B, EFloat C, EFloat *t0, EFloat *t1) {
    double discrim = (double)B.v * (double)B.v - 4. * (double)A.v * (double)C.v;
    if (discrim < 0.) return false;
    double rootDiscrim = std::sqrt(discrim);
    EFloat floatRootDiscrim((float)rootDiscrim, (float)(MachineEpsilon * rootDiscrim));
    EFloat q;
    if (B.v < 0) q = (float)-.5 * (B - floatRootDiscrim);
    else q = (float)-.5 * (B + floatRootDiscrim);
    *t0 = q / A;
    *t1 = C / q;
    if (t0->v > t1->v) std::swap(*t0, *t1);
    return true;
}

// transform.h:349-360: Transform::operator()(Ray, oError, dError)
inline Ray XfRayErr(const M44 &M, const Ray &r, V3 *oError, V3 *dError) {
    V3 o = XfPointErr(M, r.o, oError);
    V3 d = XfVectorErr(M, r.d, dError);
    Float tMax = r.tMax;
    Float lengthSquared = d.LengthSquared();
    if (lengthSquared > 0) {
        Float dt = Dot(Abs(d), *oError) / lengthSquared;
        o += d * dt;
    }
    return Ray(o, d, tMax);
}

// Sphere quadric test shared by Intersect/IntersectP
// (shapes/sphere.cpp:49-104 == :159-213).
inline bool SphereTest(const Sphere &s, const Ray &r, Ray *rayObj, V3 *pHitOut, Float *phiOut, Float *tOut) {
    Float phi; V3 pHit;
    V3 oErr, dErr;
    Ray ray = XfRayErr(s.w2o, r, &oErr, &dErr);
    EFloat ox(ray.o.x, oErr.x), oy(ray.o.y, oErr.y), oz(ray.o.z, oErr.z);
    EFloat dx(ray.d.x, dErr.x), dy(ray.d.y, dErr.y), dz(ray.d.z, dErr.z);
    EFloat a = dx * dx + dy * dy + dz * dz;
    EFloat b = 2 * (dx * ox + dy * oy + dz * oz);
    EFloat c = ox * ox + oy * oy + oz * oz - EFloat(s.radius) * EFloat(s.radius);
    EFloat t0, t1;
    if (!Quadratic(a, b, c, &t0, &t1)) return false;
    if (t0.high > ray.tMax || t1.low <= 0) return false;
    EFloat tShapeHit = t0;
    if (tShapeHit.low <= 0) {
        tShapeHit = t1;
        if (tShapeHit.high > ray.tMax) return false;
    }
    pHit = ray((Float)tShapeHit.v);
    pHit *= s.radius / Distance(pHit, V3(0, 0, 0));
    if (pHit.x == 0 && pHit.y == 0) pHit.x = 1e-5f * s.radius;
    phi = m_atan2f(pHit.y, pHit.x);
    if (phi < 0) phi += 2 * Pi;
    if ((s.zMin > -s.radius && pHit.z < s.zMin) || (s.zMax < s.radius && pHit.z > s.zMax) || phi > s.phiMax) {
        if (tShapeHit == t1) return false;
        if (t1.high > ray.tMax) return false;
        tShapeHit = t1;
        pHit = ray((Float)tShapeHit.v);
        pHit *= s.radius / Distance(pHit, V3(0, 0, 0));
        if (pHit.x == 0 && pHit.y == 0) pHit.x = 1e-5f * s.radius;
        phi = m_atan2f(pHit.y, pHit.x);
        if (phi < 0) phi += 2 * Pi;
        if ((s.zMin > -s.radius && pHit.z < s.zMin) || (s.zMax < s.radius && pHit.z > s.zMax) || phi > s.phiMax)
            return false;
    }
    *rayObj = ray; *pHitOut = pHit; *phiOut = phi; *tOut = (Float)tShapeHit.v;
    return true;
}
// shapes/sphere.cpp:49-157
inline bool SphereIntersect(const Sphere &s, bool flip, const Ray &r, Float *tHit, SurfaceInteraction *isect,
                            Counters &ctr) {
    ++ctr.sphereTests;
    Ray ray; V3 pHit; Float phi, t;
    if (!SphereTest(s, r, &ray, &pHit, &phi, &t)) return false;
    Float u = phi / s.phiMax;
    Float theta = m_acosf(Clamp(pHit.z / s.radius, -1, 1));
    Float v = (theta - s.thetaMin) / (s.thetaMax - s.thetaMin);
    Float zRadius = std::sqrt(pHit.x * pHit.x + pHit.y * pHit.y);
    Float invZRadius = 1 / zRadius;
    Float cosPhi = pHit.x * invZRadius;
    Float sinPhi = pHit.y * invZRadius;
    V3 dpdu(-s.phiMax * pHit.y, s.phiMax * pHit.x, 0);
    V3 dpdv = (s.thetaMax - s.thetaMin) * V3(pHit.z * cosPhi, pHit.z * sinPhi, -s.radius * m_sinf(theta));
    V3 pError = gamma(5) * Abs(pHit);
    SurfaceInteraction obj;
    InitSI(&obj, pHit, pError, P2(u, v), -ray.d, dpdu, dpdv, flip);
    // (*ObjectToWorld)(SurfaceInteraction), core/transform.cpp:262-297
    isect->p = XfPointErr2(s.o2w, obj.p, obj.pError, &isect->pError);
    isect->n = Normalize(XfNormal(s.w2o, obj.n));
    isect->wo = Normalize(XfVector(s.o2w, obj.wo));
    isect->uv = obj.uv;
    isect->dpdu = XfVector(s.o2w, obj.dpdu);
    isect->dpdv = XfVector(s.o2w, obj.dpdv);
    isect->shading.n = Normalize(XfNormal(s.w2o, obj.shading.n));
    isect->shading.dpdu = XfVector(s.o2w, obj.shading.dpdu);
    isect->shading.dpdv = XfVector(s.o2w, obj.shading.dpdv);
    isect->shading.n = Faceforward(isect->shading.n, isect->n);
    isect->b0 = isect->b1 = isect->b2 = 0;
    *tHit = t;
    return true;
}
inline bool SphereIntersectP(const Sphere &s, const Ray &r, Counters &ctr) {
    ++ctr.sphereTestsP;
    Ray ray; V3 pHit; Float phi, t;
    return SphereTest(s, r, &ray, &pHit, &phi, &t);
}

// ---- BVH (accelerators/bvh.cpp) -------------------------------------------
struct LinearBVHNode {          // :123-152, 32 bytes
    Float bmin[3], bmax[3];
    int32_t offset;             // leaf: primitivesOffset; interior: secondChildOffset
    uint32_t nPrimsAxis;        // (nPrimitives << 2) | axis ; axis==3 => leaf
    uint32_t nPrimitives() const { return nPrimsAxis >> 2; }
    uint32_t SplitAxis() const { return nPrimsAxis & 3; }
    bool IsLeaf() const { return (nPrimsAxis & 3) == 3; }
};
static_assert(sizeof(LinearBVHNode) == 32, "LinearBVHNode must be 32 bytes");

struct BVH {
    const Scene *scene = nullptr;
    const std::vector<PrimRef> *plist = nullptr;      // the aggregate's primitives in creation order
    const std::vector<BVH> *objects = nullptr;        // the object aggregates (for TransformedPrimitive entries)
    uint32_t orderedBase = 0;                         // position of this aggregate in the all-aggregates ordered numbering
    std::vector<LinearBVHNode> nodes;
    std::vector<uint32_t> primOrder;      // ordered index -> creation-order prim number (primNumMapping)
    int maxDepth = 0, nLeaves = 0;

    // Transform::operator()(const Bounds3f &), core/transform.cpp:238-249
    static B3 XfBounds(const M44 &M, const V3 &lo, const V3 &hi) {
        B3 ret; ret.pMin = ret.pMax = XfPoint(M, V3(lo.x, lo.y, lo.z));
        ret = Union(ret, XfPoint(M, V3(hi.x, lo.y, lo.z)));
        ret = Union(ret, XfPoint(M, V3(lo.x, hi.y, lo.z)));
        ret = Union(ret, XfPoint(M, V3(lo.x, lo.y, hi.z)));
        ret = Union(ret, XfPoint(M, V3(lo.x, hi.y, hi.z)));
        ret = Union(ret, XfPoint(M, V3(hi.x, hi.y, lo.z)));
        ret = Union(ret, XfPoint(M, V3(hi.x, lo.y, hi.z)));
        ret = Union(ret, XfPoint(M, V3(hi.x, hi.y, hi.z)));
        return ret;
    }
    B3 PrimWorldBound(uint32_t primNum) const {
        const PrimRef &pr = (*plist)[primNum];
        if (pr.shape < 0) {
            // TransformedPrimitive::WorldBound (core/primitive.h:116-118): MotionBounds of a static transform
            // is that transform applied to the wrapped primitive's bounds (BVHAccel::WorldBound, or the lone
            // primitive's own bound — the same box, a one-leaf tree's root)
            const Instance &in = scene->instances[pr.local];
            const B3 ob = (*objects)[in.object].WorldBound();
            return XfBounds(in.i2w, ob.pMin, ob.pMax);
        }
        const ShapeRec &sh = scene->shapes[pr.shape];
        if (sh.kind == SHAPE_MESH) {   // shapes/triangle.cpp:180-186
            const Mesh &m = scene->meshes[sh.meshIndex];
            const int *v = &m.idx[3 * pr.local];
            return Union(B3(m.p[v[0]], m.p[v[1]]), m.p[v[2]]);
        } else {                       // core/shape.cpp:53 + transform.cpp:237-249, sphere.cpp:43-46
            const Sphere &s = scene->spheres[sh.sphereIndex];
            return XfBounds(s.o2w, V3(-s.radius, -s.radius, s.zMin), V3(s.radius, s.radius, s.zMax));
        }
    }

    struct BuildNode {
        B3 bounds; BuildNode *children[2]; uint32_t splitAxis, firstPrimOffset, nPrimitives; bool leaf;
    };
    struct PrimInfo { size_t primitiveNumber; B3 bounds; V3 centroid; };
    struct Centroid { Float t; uint32_t primOffset, primNum; };
    struct ToDo { BuildNode *node; int start, end; };

    // accelerators/bvh.cpp:155-185 + iterativeBuild :196-333 + flatten :335-350
    void Build(const Scene *sc, const std::vector<PrimRef> *prims, const std::vector<BVH> *objs, uint32_t base) {
        scene = sc; plist = prims; objects = objs; orderedBase = base;
        const int maxPrimsInNode = smin(255, sc->prm.maxNodePrims);
        const int isectCost = sc->prm.isectCost, traversalCost = sc->prm.travCost;
        size_t n = prims->size();
        nodes.clear(); primOrder.clear();
        if (n == 0) return;
        std::vector<PrimInfo> primitiveInfo(n);
        for (size_t i = 0; i < n; ++i) {
            B3 b = PrimWorldBound((uint32_t)i);
            primitiveInfo[i].primitiveNumber = i;
            primitiveInfo[i].bounds = b;
            primitiveInfo[i].centroid = .5f * b.pMin + .5f * b.pMax;   // :59
        }
        std::vector<Centroid> centroids[3];
        for (int i = 0; i < 3; ++i) centroids[i].resize(n);
        std::vector<B3> rightToLeftBounds(n), leftToRightBounds(n);
        std::vector<BuildNode *> pool;
        auto alloc = [&]() { BuildNode *b = new BuildNode(); pool.push_back(b); return b; };
        int totalNodes = 0;
        BuildNode *root = alloc();
        std::vector<ToDo> stack;
        stack.push_back(ToDo{root, 0, (int)n});
        while (!stack.empty()) {
            ToDo cur = stack.back();
            stack.pop_back();
            totalNodes++;
            B3 bounds;
            for (int i = cur.start; i < cur.end; ++i) bounds = Union(bounds, primitiveInfo[i].bounds);
            uint32_t nPrimitives = cur.end - cur.start;
            if (nPrimitives == 1) {
                uint32_t firstPrimOffset = (uint32_t)primOrder.size();
                primOrder.push_back((uint32_t)primitiveInfo[cur.start].primitiveNumber);
                InitLeaf(cur.node, firstPrimOffset, nPrimitives, bounds);
            } else {
                uint32_t bestAxis = (uint32_t)-1, bestOffset = (uint32_t)-1, bestPrimNum = (uint32_t)-1;
                B3 bestBounds;
                Float bestCost = Infinity;
                Float oldCost = isectCost * Float(nPrimitives);
                Float totalSA = bounds.SurfaceArea();
                Float invTotalSA = 1 / totalSA;
                for (uint32_t dim = 0; dim < 3; dim++) {
                    Centroid *cd = centroids[dim].data();
                    for (uint32_t i = 0; i < nPrimitives; ++i) {
                        uint32_t pn = (uint32_t)primitiveInfo[cur.start + i].primitiveNumber;
                        cd[i] = Centroid{primitiveInfo[cur.start + i].centroid[dim], (uint32_t)(cur.start + i), pn};
                    }
                    std::sort(cd, cd + nPrimitives, [](const Centroid &e0, const Centroid &e1) -> bool {
                        if (e0.t == e1.t) return (int)e0.primNum < (int)e1.primNum;
                        else return e0.t < e1.t;
                    });
                    B3 curRL;
                    for (int i = (int)nPrimitives - 1; i >= 0; i--) {
                        curRL = Union(curRL, primitiveInfo[cd[i].primOffset].bounds);
                        rightToLeftBounds[i] = curRL;
                    }
                    B3 curLR;
                    for (int i = 0; i < (int)nPrimitives - 1; ++i) {
                        curLR = Union(curLR, primitiveInfo[cd[i].primOffset].bounds);
                        leftToRightBounds[i] = curLR;
                    }
                    for (int i = 0; i < (int)nPrimitives - 1; ++i) {
                        int primOffset = cd[i].primOffset;
                        float cost = traversalCost + isectCost *
                                                         ((i + 1) * leftToRightBounds[i].SurfaceArea() +
                                                          (nPrimitives - i - 1) * rightToLeftBounds[i + 1].SurfaceArea()) *
                                                         invTotalSA;
                        if (cost < bestCost) {
                            bestCost = cost; bestAxis = dim; bestOffset = primOffset;
                            bestPrimNum = cd[i].primNum; bestBounds = rightToLeftBounds[0];
                        }
                    }
                }
                if (bestAxis != (uint32_t)-1 && (bestCost < oldCost || nPrimitives > (uint32_t)maxPrimsInNode)) {
                    BuildNode *c0 = alloc(), *c1 = alloc();
                    const PrimInfo bestPrimitive = primitiveInfo[bestOffset];
                    const float bestCentroid = bestPrimitive.centroid[bestAxis];
                    auto pred = [&](const PrimInfo &pi) {
                        return pi.centroid[bestAxis] < bestCentroid ||
                               (pi.centroid[bestAxis] == bestCentroid && pi.primitiveNumber <= bestPrimNum);
                    };
                    // libstdc++ std::__partition, bidirectional-iterator form (stl_algo.h)
                    PrimInfo *first = &primitiveInfo[cur.start];
                    PrimInfo *last = &primitiveInfo[cur.end - 1] + 1;
                    PrimInfo *pmid;
                    while (true) {
                        while (true) {
                            if (first == last) { pmid = first; goto done; }
                            else if (pred(*first)) ++first;
                            else break;
                        }
                        --last;
                        while (true) {
                            if (first == last) { pmid = first; goto done; }
                            else if (!pred(*last)) --last;
                            else break;
                        }
                        std::swap(*first, *last);
                        ++first;
                    }
                done:
                    uint32_t mid = (uint32_t)(pmid - &primitiveInfo[0]);
                    cur.node->children[0] = c0; cur.node->children[1] = c1;
                    cur.node->splitAxis = bestAxis; cur.node->nPrimitives = nPrimitives;
                    cur.node->bounds = bestBounds; cur.node->leaf = false;
                    stack.push_back(ToDo{c0, cur.start, (int)mid});
                    stack.push_back(ToDo{c1, (int)mid, cur.end});
                } else {
                    uint32_t firstPrimOffset = (uint32_t)primOrder.size();
                    for (int i = cur.start; i < cur.end; ++i)
                        primOrder.push_back((uint32_t)primitiveInfo[i].primitiveNumber);
                    InitLeaf(cur.node, firstPrimOffset, nPrimitives, bounds);
                }
            }
        }
        nodes.resize(totalNodes);
        int offset = 0;
        maxDepth = 0; nLeaves = 0;
        Flatten(root, &offset, 1);
        for (BuildNode *b : pool) delete b;
    }
    static void InitLeaf(BuildNode *nd, uint32_t first, uint32_t n, const B3 &b) {
        nd->firstPrimOffset = first; nd->nPrimitives = n; nd->bounds = b;
        nd->children[0] = nd->children[1] = nullptr; nd->leaf = true;
    }
    int Flatten(BuildNode *node, int *offset, int depth) {
        LinearBVHNode *ln = &nodes[*offset];
        int myOffset = (*offset)++;
        if (depth > maxDepth) maxDepth = depth;
        for (int k = 0; k < 3; ++k) { ln->bmin[k] = node->bounds.pMin[k]; ln->bmax[k] = node->bounds.pMax[k]; }
        if (node->leaf) {
            ++nLeaves;
            ln->nPrimsAxis = 3u | (node->nPrimitives << 2);
            ln->offset = (int32_t)node->firstPrimOffset;
        } else {
            ln->nPrimsAxis = node->splitAxis | (node->nPrimitives << 2);
            Flatten(node->children[0], offset, depth + 1);
            ln->offset = Flatten(node->children[1], offset, depth + 1);
        }
        return myOffset;
    }

    // core/geometry.h:1754-1780
    static bool SlabTest(const LinearBVHNode *nd, const Ray &ray, const V3 &invDir, const int dirIsNeg[3]) {
        const Float *b[2] = {nd->bmin, nd->bmax};
        Float tMin = (b[dirIsNeg[0]][0] - ray.o.x) * invDir.x;
        Float tMax = (b[1 - dirIsNeg[0]][0] - ray.o.x) * invDir.x;
        Float tyMin = (b[dirIsNeg[1]][1] - ray.o.y) * invDir.y;
        Float tyMax = (b[1 - dirIsNeg[1]][1] - ray.o.y) * invDir.y;
        tMax *= 1 + 2 * gamma(3);
        tyMax *= 1 + 2 * gamma(3);
        if (tMin > tyMax || tyMin > tMax) return false;
        if (tyMin > tMin) tMin = tyMin;
        if (tyMax < tMax) tMax = tyMax;
        Float tzMin = (b[dirIsNeg[2]][2] - ray.o.z) * invDir.z;
        Float tzMax = (b[1 - dirIsNeg[2]][2] - ray.o.z) * invDir.z;
        tzMax *= 1 + 2 * gamma(3);
        if (tMin > tzMax || tzMin > tMax) return false;
        if (tzMin > tMin) tMin = tzMin;
        if (tzMax < tMax) tMax = tzMax;
        return (tMin < ray.tMax) && (tMax > 0);
    }

    // TransformedPrimitive::Intersect / IntersectP, core/primitive.cpp:77-102 (static transform:
    // AnimatedTransform::Interpolate returns the start transform)
    bool InstanceIntersect(int instIndex, const Ray &r, SurfaceInteraction *isect, Counters &ctr) const {
        const Instance &in = scene->instances[instIndex];
        Ray ray = XfRay(in.w2i, r);
        const BVH &ob = (*objects)[in.object];
        // ObjectInstance wraps an accelerator only around more than one primitive (core/api.cpp:1798-1806)
        const bool intersects = ob.plist->size() > 1 ? ob.Intersect(ray, isect, ctr) : ob.PrimIntersect(0, ray, isect, ctr);
        if (!intersects) return false;
        r.tMax = ray.tMax;
        if (!IsIdentity(in.i2w)) {
            // Transform::operator()(const SurfaceInteraction &), core/transform.cpp:262-297
            SurfaceInteraction si = *isect;
            isect->p = XfPointErr2(in.i2w, si.p, si.pError, &isect->pError);
            isect->n = Normalize(XfNormal(in.w2i, si.n));
            isect->wo = Normalize(XfVector(in.i2w, si.wo));
            isect->dpdu = XfVector(in.i2w, si.dpdu);
            isect->dpdv = XfVector(in.i2w, si.dpdv);
            isect->shading.n = Normalize(XfNormal(in.w2i, si.shading.n));
            isect->shading.dpdu = XfVector(in.i2w, si.shading.dpdu);
            isect->shading.dpdv = XfVector(in.i2w, si.shading.dpdv);
            isect->shading.n = Faceforward(isect->shading.n, isect->n);
        }
        isect->inst = instIndex;
        return true;
    }
    bool InstanceIntersectP(int instIndex, const Ray &r, Counters &ctr) const {
        const Instance &in = scene->instances[instIndex];
        Ray ray = XfRay(in.w2i, r);
        const BVH &ob = (*objects)[in.object];
        return ob.plist->size() > 1 ? ob.IntersectP(ray, ctr) : ob.PrimIntersectP(0, ray, ctr);
    }

    // GeometricPrimitive::Intersect, core/primitive.cpp:123-138
    bool PrimIntersect(uint32_t ordered, const Ray &r, SurfaceInteraction *isect, Counters &ctr) const {
        uint32_t primNum = primOrder[ordered];
        const PrimRef &pr = (*plist)[primNum];
        if (pr.shape < 0) return InstanceIntersect(pr.local, r, isect, ctr);
        const ShapeRec &sh = scene->shapes[pr.shape];
        bool flip = (sh.reverseOrientation != 0) ^ (sh.swapsHandedness != 0);
        Float tHit;
        if (sh.kind == SHAPE_MESH) {
            const Mesh &m = scene->meshes[sh.meshIndex];
            TriRef tr{&m, &m.idx[3 * pr.local], flip};
            if (!TriangleIntersect(tr, r, &tHit, isect, ctr)) return false;
            isect->tri = pr.local;
        } else {
            if (!SphereIntersect(scene->spheres[sh.sphereIndex], flip, r, &tHit, isect, ctr)) return false;
            isect->tri = -1;
        }
        r.tMax = tHit;
        isect->prim = (int)primNum;
        isect->shape = pr.shape;
        isect->ordered = (int)(orderedBase + ordered);
        isect->inst = -1;
        return true;
    }
    bool PrimIntersectP(uint32_t ordered, const Ray &r, Counters &ctr) const {
        uint32_t primNum = primOrder[ordered];
        const PrimRef &pr = (*plist)[primNum];
        if (pr.shape < 0) return InstanceIntersectP(pr.local, r, ctr);
        const ShapeRec &sh = scene->shapes[pr.shape];
        if (sh.kind == SHAPE_MESH) {
            const Mesh &m = scene->meshes[sh.meshIndex];
            TriRef tr{&m, &m.idx[3 * pr.local], false};
            return TriangleIntersectP(tr, r, ctr);
        } else return SphereIntersectP(scene->spheres[sh.sphereIndex], r, ctr);
    }

    // accelerators/bvh.cpp:354-396
    // optional per-fetch trace (0 = node step without entering a leaf, k = leaf entered with k prims); study aid
    mutable std::vector<uint8_t> *fetchTrace = nullptr;
    bool Intersect(const Ray &ray, SurfaceInteraction *isect, Counters &ctr, int *orderedHit = nullptr) const {
        if (nodes.empty()) return false;
        bool hit = false;
        V3 invDir(1 / ray.d.x, 1 / ray.d.y, 1 / ray.d.z);
        int dirIsNeg[3] = {invDir.x < 0, invDir.y < 0, invDir.z < 0};
        int toVisitOffset = 0, currentNodeIndex = 0;
        int nodesToVisit[64];
        while (true) {
            const LinearBVHNode *node = &nodes[currentNodeIndex];
            ++ctr.nodesFetched;
            const bool slabHit = SlabTest(node, ray, invDir, dirIsNeg);
            if (fetchTrace) fetchTrace->push_back((slabHit && node->IsLeaf()) ? (uint8_t)node->nPrimitives() : (uint8_t)0);
            if (slabHit) {
                ++ctr.nodesEntered;
                if (node->IsLeaf()) {
                    ++ctr.leavesEntered;
                    for (uint32_t i = 0; i < node->nPrimitives(); ++i)
                        if (PrimIntersect(node->offset + i, ray, isect, ctr)) {
                            hit = true;
                            if (orderedHit) *orderedHit = isect->ordered;
                        }
                    if (toVisitOffset == 0) break;
                    currentNodeIndex = nodesToVisit[--toVisitOffset];
                } else {
                    if (dirIsNeg[node->SplitAxis()]) {
                        nodesToVisit[toVisitOffset++] = currentNodeIndex + 1;
                        currentNodeIndex = node->offset;
                    } else {
                        nodesToVisit[toVisitOffset++] = node->offset;
                        currentNodeIndex = currentNodeIndex + 1;
                    }
                }
            } else {
                if (toVisitOffset == 0) break;
                currentNodeIndex = nodesToVisit[--toVisitOffset];
            }
        }
        return hit;
    }
    // accelerators/bvh.cpp:398-437
    bool IntersectP(const Ray &ray, Counters &ctr) const {
        if (nodes.empty()) return false;
        V3 invDir(1.f / ray.d.x, 1.f / ray.d.y, 1.f / ray.d.z);
        int dirIsNeg[3] = {invDir.x < 0, invDir.y < 0, invDir.z < 0};
        int nodesToVisit[64];
        int toVisitOffset = 0, currentNodeIndex = 0;
        while (true) {
            const LinearBVHNode *node = &nodes[currentNodeIndex];
            ++ctr.nodesFetchedP;
            if (SlabTest(node, ray, invDir, dirIsNeg)) {
                ++ctr.nodesEnteredP;
                if (node->IsLeaf()) {
                    ++ctr.leavesEnteredP;
                    for (uint32_t i = 0; i < node->nPrimitives(); ++i)
                        if (PrimIntersectP(node->offset + i, ray, ctr)) return true;
                    if (toVisitOffset == 0) break;
                    currentNodeIndex = nodesToVisit[--toVisitOffset];
                } else {
                    if (dirIsNeg[node->SplitAxis()]) {
                        nodesToVisit[toVisitOffset++] = currentNodeIndex + 1;
                        currentNodeIndex = node->offset;
                    } else {
                        nodesToVisit[toVisitOffset++] = node->offset;
                        currentNodeIndex = currentNodeIndex + 1;
                    }
                }
            } else {
                if (toVisitOffset == 0) break;
                currentNodeIndex = nodesToVisit[--toVisitOffset];
            }
        }
        return false;
    }
    B3 WorldBound() const {
        B3 b;
        if (!nodes.empty()) {
            b.pMin = V3(nodes[0].bmin[0], nodes[0].bmin[1], nodes[0].bmin[2]);
            b.pMax = V3(nodes[0].bmax[0], nodes[0].bmax[1], nodes[0].bmax[2]);
        }
        return b;
    }
};

}  // namespace orc
