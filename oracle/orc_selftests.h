// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
// The reference's own unit tests for this path, restated over the oracle's
// functions so that they pin the restatement:
//   Triangle.Watertight      src/tests/shapes.cpp:28-129
//   Triangle.Reintersect     src/tests/shapes.cpp:154-205
//   LowDiscrepancy.RadicalInverse / ScrambledRadicalInverse   src/tests/sampling.cpp:15-74
// Each returns the number of failed expectations (0 == pass).
#pragma once
#include "orc_integrator.h"

namespace orc {

inline int SelfTestWatertight(int nRays) {
    RNG rng(12111);
    const int nTheta = 16, nPhi = 16;
    Mesh mesh; mesh.hasN = mesh.hasUV = mesh.hasS = false;
    std::vector<V3> &vertices = mesh.p;
    for (int t = 0; t < nTheta; ++t) {
        Float theta = Pi * (Float)t / (Float)(nTheta - 1);
        Float cosTheta = std::cos(theta), sinTheta = std::sin(theta);
        for (int p = 0; p < nPhi; ++p) {
            Float phi = 2 * Pi * (Float)p / (Float)(nPhi - 1);
            Float radius = 1;
            if (t == 0) vertices.push_back(V3(0, 0, radius));
            else if (t == nTheta - 1) vertices.push_back(V3(0, 0, -radius));
            else if (p == nPhi - 1) vertices.push_back(vertices[vertices.size() - (nPhi - 1)]);
            else {
                radius += 5 * rng.UniformFloat();
                vertices.push_back(V3(0, 0, 0) + radius * V3(sinTheta * std::cos(phi), sinTheta * std::sin(phi), cosTheta));
            }
        }
    }
    std::vector<int> &indices = mesh.idx;
    auto offset = [nPhi](int t, int p) { return t * nPhi + p; };
    for (int p = 0; p < nPhi - 1; ++p) { indices.push_back(offset(0, 0)); indices.push_back(offset(1, p)); indices.push_back(offset(1, p + 1)); }
    for (int t = 1; t < nTheta - 2; ++t)
        for (int p = 0; p < nPhi - 1; ++p) {
            indices.push_back(offset(t, p)); indices.push_back(offset(t + 1, p)); indices.push_back(offset(t + 1, p + 1));
            indices.push_back(offset(t, p)); indices.push_back(offset(t + 1, p + 1)); indices.push_back(offset(t, p + 1));
        }
    for (int p = 0; p < nPhi - 1; ++p) { indices.push_back(offset(nTheta - 1, 0)); indices.push_back(offset(nTheta - 2, p)); indices.push_back(offset(nTheta - 2, p + 1)); }
    mesh.nTris = (uint32_t)(indices.size() / 3); mesh.nVerts = (uint32_t)vertices.size();
    int failures = 0;
    Counters ctr;
    auto countHits = [&](const Ray &r) {
        int nHits = 0;
        for (uint32_t i = 0; i < mesh.nTris; ++i) {
            TriRef tr{&mesh, &mesh.idx[3 * i], false};
            Float tHit; SurfaceInteraction isect;
            Ray rr(r.o, r.d, r.tMax);      // every triangle is tested against the unshortened ray, as in the reference test
            if (TriangleIntersect(tr, rr, &tHit, &isect, ctr)) ++nHits;
        }
        return nHits;
    };
    for (int i = 0; i < nRays; ++i) {
        RNG rr(i);
        P2 u; u.x = rr.UniformFloat(); u.y = rr.UniformFloat();
        V3 p = V3(0, 0, 0) + Float(0.5) * UniformSampleSphere(u);
        u.x = rr.UniformFloat(); u.y = rr.UniformFloat();
        Ray r(p, UniformSampleSphere(u));
        if (countHits(r) < 1) ++failures;
        V3 pVertex = vertices[rr.UniformUInt32((uint32_t)vertices.size())];
        r.d = pVertex - r.o;
        if (countHits(r) < 1) ++failures;
    }
    return failures;
}

inline Float pExp(RNG &rng, Float e = 8.) { Float logu = Lerp(rng.UniformFloat(), -e, e); return std::pow((Float)10, logu); }

inline int SelfTestReintersect(int nTriangles, int nRaysPerTriangle, int *nTested) {
    int failures = 0, tested = 0;
    Counters ctr;
    for (int i = 0; i < nTriangles; ++i) {
        RNG rng(i);
        Mesh mesh; mesh.hasN = mesh.hasUV = mesh.hasS = false; mesh.nTris = 1; mesh.nVerts = 3; mesh.idx = {0, 1, 2};
        V3 v[3];
        for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) v[j][k] = pExp(rng);
        if (Cross(v[1] - v[0], v[2] - v[0]).LengthSquared() < 1e-20) continue;
        mesh.p = {v[0], v[1], v[2]};
        TriRef tr{&mesh, mesh.idx.data(), false};
        // Triangle::Sample(u), shapes/triangle.cpp:596-621 (point only)
        P2 u; u.x = rng.UniformFloat(); u.y = rng.UniformFloat();
        Float su0 = std::sqrt(u.x); Float b0 = 1 - su0, b1 = u.y * su0;
        V3 pTri = b0 * v[0] + b1 * v[1] + (1 - b0 - b1) * v[2];
        V3 o; for (int j = 0; j < 3; ++j) o[j] = pExp(rng);
        Ray r(o, pTri - o);
        Float tHit; SurfaceInteraction isect;
        if (!TriangleIntersect(tr, r, &tHit, &isect, ctr)) continue;
        ++tested;
        for (int j = 0; j < nRaysPerTriangle; ++j) {
            P2 uu; uu.x = rng.UniformFloat(); uu.y = rng.UniformFloat();
            V3 w = UniformSampleSphere(uu);
            Ray rOut = SpawnRay(isect.p, isect.pError, isect.n, w);
            if (TriangleIntersectP(tr, rOut, ctr)) ++failures;
            SurfaceInteraction s2; Float t2;
            if (TriangleIntersect(tr, rOut, &t2, &s2, ctr)) ++failures;
            V3 p2; for (int k = 0; k < 3; ++k) p2[k] = pExp(rng);
            // Interaction::SpawnRayTo(const Point3f&), core/interaction.h:68-72
            V3 origin = OffsetRayOrigin(isect.p, isect.pError, isect.n, p2 - isect.p);
            Ray rTo(origin, p2 - isect.p, 1 - ShadowEpsilon);
            if (TriangleIntersectP(tr, rTo, ctr)) ++failures;
            if (TriangleIntersect(tr, rTo, &t2, &s2, ctr)) ++failures;
        }
    }
    if (nTested) *nTested = tested;
    return failures;
}

inline int SelfTestRadicalInverse() {
    int failures = 0;
    for (int a = 0; a < 1024; ++a)
        if (ReverseBits32(a) * 2.3283064365386963e-10f != RadicalInverse(0, a)) ++failures;
    return failures;
}

inline int SelfTestScrambledRadicalInverse() {
    int failures = 0;
    for (int dim = 0; dim < 128; ++dim) {
        RNG rng(dim);
        const int base = Primes()[dim];
        std::vector<uint16_t> perm;
        for (int i = 0; i < base; ++i) perm.push_back(base - 1 - i);
        for (int k = 0; k < base; ++k) { int other = k + rng.UniformUInt32(base - k); std::swap(perm[k], perm[other]); }
        const uint32_t idxs[7] = {0, 1, 2, 1151, 32351, 4363211, 681122};
        for (uint32_t index : idxs) {
            {   // pbrt-v2 style evaluation
                Float val = 0;
                Float invBase = 1. / base, invBi = invBase;
                uint32_t n = index;
                while (n > 0) { uint32_t d_i = perm[n % base]; val += d_i * invBi; n *= invBase; invBi *= invBase; }
                val += perm[0] * base / (base - 1.0f) * invBi;
                if (!(std::abs(val - ScrambledRadicalInverse(dim, index, &perm[0])) <= 1e-5)) ++failures;
            }
            {   // naive 32-digit loop
                Float val = 0;
                Float invBase = 1. / base, invBi = invBase;
                uint32_t a = index;
                for (int i = 0; i < 32; ++i) { uint32_t d_i = perm[a % base]; a /= base; val += d_i * invBi; invBi *= invBase; }
                if (!(std::abs(val - ScrambledRadicalInverse(dim, index, &perm[0])) <= 1e-5)) ++failures;
            }
        }
    }
    return failures;
}

// The claim behind the product's quadric pre-test (csrc/device/dev_intersect.h: sphere_may_hit): evaluating only the VALUE
// lanes of SphereTest's EFloat arithmetic, "discriminant < 0", "t0.v > tMax" and "t1.v <= 0" each imply that the full
// interval test returns false.  Counts rays the value-lane test rejects although SphereTest accepts them (must be 0), and
// reports how many rays each side accepted.  Rays: origins around the sphere at every scale, directions partly aimed at it,
// tMax partly just short of / just beyond the surface (the shadow-ray situation).
inline int SelfTestSpherePretest(int nRays, int *nFull, int *nMaybe) {
    int violations = 0, full = 0, maybe = 0;
    RNG rng(77);
    for (int i = 0; i < nRays; ++i) {
        Sphere s;
        const Float radius = pExp(rng, 2.);
        V3 c(Lerp(rng.UniformFloat(), -50, 50), Lerp(rng.UniformFloat(), -50, 50), Lerp(rng.UniformFloat(), -50, 50));
        const Float sx = pExp(rng, .5), sy = pExp(rng, .5), sz = pExp(rng, .5);
        Xf x = XfMul(XfTranslate(c), XfScale(sx, sy, sz));
        s.o2w = x.m; s.w2o = x.mInv;
        s.radius = radius; s.zMin = -radius; s.zMax = radius; s.thetaMin = Pi; s.thetaMax = 0; s.phiMax = 2 * Pi;
        if (i % 7 == 0) { s.zMax = radius * 0.4f; s.phiMax = 4.f; }
        V3 o = c + V3(Lerp(rng.UniformFloat(), -4, 4) * radius * sx, Lerp(rng.UniformFloat(), -4, 4) * radius * sy, Lerp(rng.UniformFloat(), -4, 4) * radius * sz);
        V3 target = c + V3(Lerp(rng.UniformFloat(), -1.3f, 1.3f) * radius * sx, Lerp(rng.UniformFloat(), -1.3f, 1.3f) * radius * sy, Lerp(rng.UniformFloat(), -1.3f, 1.3f) * radius * sz);
        V3 d = target - o;
        if (i % 3 == 0) d = Normalize(d);
        Float tMax = Infinity;
        Ray probe(o, d, Infinity);
        Ray ro; V3 ph; Float phi, t;
        if (i % 2 == 0 && SphereTest(s, probe, &ro, &ph, &phi, &t)) {
            const Float f[5] = {0.9999f, 0.99999994f, 1.f, 1.0000001f, 1.0001f};
            tMax = t * f[i % 5];
        }
        Ray r(o, d, tMax);
        // value lanes only
        V3 oErr, dErr;
        Ray ray = XfRayErr(s.w2o, r, &oErr, &dErr);
        const Float a = (ray.d.x * ray.d.x + ray.d.y * ray.d.y) + ray.d.z * ray.d.z;
        const Float b = 2.f * ((ray.d.x * ray.o.x + ray.d.y * ray.o.y) + ray.d.z * ray.o.z);
        const Float cc = ((ray.o.x * ray.o.x + ray.o.y * ray.o.y) + ray.o.z * ray.o.z) - s.radius * s.radius;
        bool may = true;
        const double discrim = (double)b * (double)b - 4. * (double)a * (double)cc;
        if (discrim < 0.) may = false;
        else {
            const Float fr = (Float)std::sqrt(discrim);
            const Float q = b < 0 ? -.5f * (b - fr) : -.5f * (b + fr);
            Float t0 = q / a, t1 = cc / q;
            if (t0 > t1) std::swap(t0, t1);
            if (t0 > r.tMax || t1 <= 0) may = false;
        }
        const bool hit = SphereTest(s, r, &ro, &ph, &phi, &t);
        if (hit) ++full;
        if (may) ++maybe;
        if (hit && !may) ++violations;
    }
    *nFull = full; *nMaybe = maybe;
    return violations;
}

}  // namespace orc
