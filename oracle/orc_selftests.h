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

// The jittered-sphere mesh of Triangle.Watertight (src/tests/shapes.cpp:28-92): 16 x 16 vertices, 420 triangles
inline void MakeWatertightMesh(Mesh &mesh) {
    RNG rng(12111);
    const int nTheta = 16, nPhi = 16;
    mesh.hasN = mesh.hasUV = mesh.hasS = false;
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
}
// The two rays of iteration i of that test (:94-128): from a random point within radius 0.5 in a random direction, then
// straight at a random vertex
inline void WatertightRays(const Mesh &mesh, int i, Ray *a, Ray *b) {
    RNG rr(i);
    P2 u; u.x = rr.UniformFloat(); u.y = rr.UniformFloat();
    V3 p = V3(0, 0, 0) + Float(0.5) * UniformSampleSphere(u);
    u.x = rr.UniformFloat(); u.y = rr.UniformFloat();
    *a = Ray(p, UniformSampleSphere(u));
    V3 pVertex = mesh.p[rr.UniformUInt32((uint32_t)mesh.p.size())];
    *b = Ray(p, pVertex - p);
}

inline int SelfTestWatertight(int nRays) {
    Mesh mesh;
    MakeWatertightMesh(mesh);
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
        Ray a, b;
        WatertightRays(mesh, i, &a, &b);
        if (countHits(a) < 1) ++failures;
        if (countHits(b) < 1) ++failures;
    }
    return failures;
}
// The same rays for a batched closest-hit query (the GPU test sends them through hprt_intersect): rays 2i and 2i + 1 of
// iteration i, and the closest hit distance over all triangles by brute force (tMax shrinking as hits are accepted: the
// minimum whatever the order), -1 where nothing is hit.
inline void WatertightCase(int nIter, float *P, int *idx, float *o, float *d, float *tBrute) {
    Mesh mesh;
    MakeWatertightMesh(mesh);
    if (P) for (size_t i = 0; i < mesh.p.size(); ++i) { P[3 * i] = mesh.p[i].x; P[3 * i + 1] = mesh.p[i].y; P[3 * i + 2] = mesh.p[i].z; }
    if (idx) for (size_t i = 0; i < mesh.idx.size(); ++i) idx[i] = mesh.idx[i];
    Counters ctr;
    for (int i = 0; i < nIter; ++i) {
        Ray r[2];
        WatertightRays(mesh, i, &r[0], &r[1]);
        for (int k = 0; k < 2; ++k) {
            const size_t j = 2 * (size_t)i + k;
            o[3 * j] = r[k].o.x; o[3 * j + 1] = r[k].o.y; o[3 * j + 2] = r[k].o.z;
            d[3 * j] = r[k].d.x; d[3 * j + 1] = r[k].d.y; d[3 * j + 2] = r[k].d.z;
            Ray rr(r[k].o, r[k].d, Infinity);
            bool any = false;
            for (uint32_t t = 0; t < mesh.nTris; ++t) {
                TriRef tr{&mesh, &mesh.idx[3 * t], false};
                Float tHit; SurfaceInteraction isect;
                if (TriangleIntersect(tr, rr, &tHit, &isect, ctr)) { rr.tMax = tHit; any = true; }
            }
            tBrute[j] = any ? rr.tMax : -1.f;
        }
    }
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

// ---- FullSphere.Reintersect / PartialSphere.Reintersect (src/tests/shapes.cpp:374-425, 427-441, 481-500) ----
// One case = one random sphere, one ray that hits it, and 2 * nRays rays leaving the hit point on the outer side (SpawnRay
// in a random direction of the normal's hemisphere, SpawnRayTo a random point of that hemisphere): none may hit the sphere
// again, neither in IntersectP nor in Intersect.  rays (may be null): 7 floats per ray {o, d, tMax}.  Returns false when the
// first ray misses (the reference test then skips the case).
inline bool SphereReintersectCase(int seed, bool partial, int nRays, Sphere *sphere, Ray *first, Float *tFirst, float *rays, int *failures) {
    RNG rng(seed);
    const Float radius = pExp(rng, 4);
    Float zMin = -radius, zMax = radius, phiMaxDeg = 360;
    if (partial) {
        zMin = rng.UniformFloat() < 0.5 ? -radius : Lerp(rng.UniformFloat(), -radius, radius);
        zMax = rng.UniformFloat() < 0.5 ? radius : Lerp(rng.UniformFloat(), -radius, radius);
        phiMaxDeg = rng.UniformFloat() < 0.5 ? 360. : rng.UniformFloat() * 360.;
    }
    Sphere s;      // Sphere::Sphere, shapes/sphere.h:50-59
    Xf id = XfTranslate(V3(0, 0, 0));
    s.o2w = id.m; s.w2o = id.mInv;
    s.radius = radius;
    s.zMin = Clamp(smin(zMin, zMax), -radius, radius);
    s.zMax = Clamp(smax(zMin, zMax), -radius, radius);
    s.thetaMin = std::acos(Clamp(smin(zMin, zMax) / radius, -1, 1));
    s.thetaMax = std::acos(Clamp(smax(zMin, zMax) / radius, -1, 1));
    s.phiMax = Radians(Clamp(phiMaxDeg, 0, 360));
    *sphere = s;
    // TestReintersectConvex
    V3 o; for (int c = 0; c < 3; ++c) o[c] = pExp(rng);
    const V3 bmin(-radius, -radius, s.zMin), bmax(radius, radius, s.zMax);      // Sphere::ObjectBound under the identity
    V3 t; for (int c = 0; c < 3; ++c) t[c] = rng.UniformFloat();
    V3 p2(Lerp(t.x, bmin.x, bmax.x), Lerp(t.y, bmin.y, bmax.y), Lerp(t.z, bmin.z, bmax.z));
    Ray r(o, p2 - o);
    if (rng.UniformFloat() < .5) r.d = Normalize(r.d);
    *first = r;
    Counters ctr;
    SurfaceInteraction isect; Float tHit;
    if (!SphereIntersect(s, false, r, &tHit, &isect, ctr)) return false;
    *tFirst = tHit;
    int fails = 0;
    auto emit = [&](int j, const Ray &q) {
        if (rays) { float *w = rays + 7 * (size_t)j; w[0] = q.o.x; w[1] = q.o.y; w[2] = q.o.z; w[3] = q.d.x; w[4] = q.d.y; w[5] = q.d.z; w[6] = q.tMax; }
        if (SphereIntersectP(s, q, ctr)) ++fails;
        Ray q2(q.o, q.d, q.tMax);
        SurfaceInteraction s2; Float t2;
        if (SphereIntersect(s, false, q2, &t2, &s2, ctr)) ++fails;
    };
    for (int j = 0; j < nRays; ++j) {
        P2 u; u.x = rng.UniformFloat(); u.y = rng.UniformFloat();
        V3 w = UniformSampleSphere(u);
        w = Faceforward(w, isect.n);
        emit(2 * j, SpawnRay(isect.p, isect.pError, isect.n, w));
        V3 q; for (int c = 0; c < 3; ++c) q[c] = pExp(rng);
        w = q - isect.p;
        w = Faceforward(w, isect.n);
        q = isect.p + w;
        // Interaction::SpawnRayTo(const Point3f &), core/interaction.h:68-72
        emit(2 * j + 1, Ray(OffsetRayOrigin(isect.p, isect.pError, isect.n, q - isect.p), q - isect.p, 1 - ShadowEpsilon));
    }
    *failures = fails;
    return true;
}
inline int SelfTestSphereReintersect(int nSpheres, int nRays, int *nTested) {
    int failures = 0, tested = 0;
    for (int partial = 0; partial < 2; ++partial)
        for (int i = 0; i < nSpheres; ++i) {
            Sphere s; Ray first; Float t; int f = 0;
            if (!SphereReintersectCase(i, partial != 0, nRays, &s, &first, &t, nullptr, &f)) continue;
            ++tested; failures += f;
        }
    if (nTested) *nTested = tested;
    return failures;
}

// ---- FloatingPoint.NextUpDownFloat (src/tests/fp_tests.cpp:29-47) ----
inline int SelfTestNextFloat() {
    int failures = 0;
    if (!(NextFloatUp(-0.f) > 0.f)) ++failures;
    if (!(NextFloatDown(0.f) < 0.f)) ++failures;
    if (!(NextFloatUp((float)Infinity) == (float)Infinity)) ++failures;
    if (!(NextFloatDown((float)Infinity) < (float)Infinity)) ++failures;
    if (!(NextFloatDown(-(float)Infinity) == -(float)Infinity)) ++failures;
    if (!(NextFloatUp(-(float)Infinity) > -(float)Infinity)) ++failures;
    RNG rng;
    for (int i = 0; i < 100000; ++i) {
        float f;
        do { uint32_t b = rng.UniformUInt32(); memcpy(&f, &b, 4); } while (std::isnan(f));
        if (std::isinf(f)) continue;
        if (std::nextafter(f, (float)Infinity) != NextFloatUp(f)) ++failures;
        if (std::nextafter(f, -(float)Infinity) != NextFloatDown(f)) ++failures;
    }
    return failures;
}

// ---- EFloat.Add / Sub / Mul / Div (src/tests/fp_tests.cpp:107-262): the interval of the result contains the result of the
// precise values.  (Abs and Sqrt are not on the path: Sphere::Intersect uses +, -, *, / and the double-precision Quadratic.) ----
inline EFloat SelfTestGetEFloat(RNG &rng, Float minExp = -6., Float maxExp = 6.) {
    Float logu = Lerp(rng.UniformFloat(), minExp, maxExp);
    Float val = std::pow((Float)10, logu);
    Float err = 0;
    switch (rng.UniformUInt32(4)) {
    case 0: break;
    case 1: { uint32_t ulpError = rng.UniformUInt32(1024); uint32_t b; memcpy(&b, &val, 4); b += ulpError; Float offset; memcpy(&offset, &b, 4); err = std::abs(offset - val); break; }
    case 2: { uint32_t ulpError = rng.UniformUInt32(1024 * 1024); uint32_t b; memcpy(&b, &val, 4); b += ulpError; Float offset; memcpy(&offset, &b, 4); err = std::abs(offset - val); break; }
    case 3: err = (4 * rng.UniformFloat()) * std::abs(val);
    }
    Float sign = rng.UniformFloat() < .5 ? -1. : 1.;
    return EFloat(sign * val, err);
}
inline double SelfTestGetPrecise(const EFloat &ef, RNG &rng) {
    switch (rng.UniformUInt32(3)) {
    case 0: return ef.low;
    case 1: return ef.high;
    case 2: {
        Float t = rng.UniformFloat();
        double p = (1 - t) * ef.low + t * ef.high;
        if (p > ef.high) p = ef.high;
        if (p < ef.low) p = ef.low;
        return p;
    }
    }
    return ef.v;
}
inline int SelfTestEFloat(int iters) {
    int failures = 0;
    for (int op = 0; op < 4; ++op)
        for (int trial = 0; trial < iters; ++trial) {
            RNG rng(trial);
            EFloat ef[2]; ef[0] = SelfTestGetEFloat(rng); ef[1] = SelfTestGetEFloat(rng);
            double precise[2]; precise[0] = SelfTestGetPrecise(ef[0], rng); precise[1] = SelfTestGetPrecise(ef[1], rng);
            EFloat r; float pr;
            if (op == 0) { r = ef[0] + ef[1]; pr = precise[0] + precise[1]; }
            else if (op == 1) { r = ef[0] - ef[1]; pr = precise[0] - precise[1]; }
            else if (op == 2) { r = ef[0] * ef[1]; pr = precise[0] * precise[1]; }
            else {
                // (err = high - low of the EFloat: GetAbsoluteError, core/efloat.h:98)
                if ((double)ef[1].low * (double)ef[1].high < 0. || (ef[1].high - ef[1].low) > .25 * std::abs(ef[1].low)) continue;
                r = ef[0] / ef[1]; pr = precise[0] / precise[1];
            }
            if (!(pr >= r.low)) ++failures;
            if (!(pr <= r.high)) ++failures;
        }
    return failures;
}

// ---- Distribution1D.Discrete (src/tests/sampling.cpp:231-282), the part the path uses (SampleDiscrete with its pdf) ----
inline int SelfTestDistribution1D() {
    int failures = 0;
    Float func[4] = {0, 1., 0., 3.};
    Distribution1D dist; dist.Init(func, 4);
    if (dist.Count() != 4) ++failures;
    auto discretePdf = [&](int i) { return dist.func[i] / (dist.funcInt * dist.Count()); };      // core/sampling.h:97-100
    if (discretePdf(0) != 0 || discretePdf(1) != .25f || discretePdf(2) != 0 || discretePdf(3) != .75f) ++failures;
    Float pdf;
    const Float us[7] = {0.f, 0.125f, .24999f, .250001f, 0.625f, OneMinusEpsilon, 1.f};
    const int want[7] = {1, 1, 1, 3, 3, 3, 3};
    for (int k = 0; k < 7; ++k) {
        if (dist.SampleDiscrete(us[k], &pdf) != want[k]) ++failures;
        if (pdf != (want[k] == 1 ? 0.25f : 0.75f)) ++failures;
    }
    Float u = .25, uMax = .25;
    for (int i = 0; i < 20; ++i) { u = NextFloatDown(u); uMax = NextFloatUp(uMax); }
    for (; u < uMax; u = NextFloatUp(u)) {
        int interval = dist.SampleDiscrete(u, &pdf);
        if (interval == 3) break;
        if (interval != 1) ++failures;
    }
    if (!(u < uMax)) ++failures;
    for (; u <= uMax; u = NextFloatUp(u))
        if (dist.SampleDiscrete(u, &pdf) != 3) ++failures;
    return failures;
}


// ---------------------------------------------------------------------------
// BSDFSampling.* of src/tests/bsdfs.cpp: the chi-square test that a BxDF's Sample_f() draws directions with the density its
// Pdf() reports — a 10 x 20 histogram over (theta, phi) of a million sampled directions against the density integrated over
// each cell by nested adaptive Simpson quadrature, five random outgoing directions, significance 0.01 with the Sidak
// correction.  Restated for the cases inside the path's scope (the visible-area Trowbridge-Reitz sampling the reference uses by
// default): Lambertian (:484), TR_VA_0p5 (:492-496), TR_VA_0p3_0p15 (:516-520), TR_VA_0p3 = FresnelBlend (:540-544).  The
// reference's microfacet cases use FresnelNoOp; the oracle's microfacet lobe carries plastic's dielectric Fresnel, which
// changes f() only by a factor that is never zero, and the test reads f() only to skip zero samples (:190).
// RNG: the reference's default-seeded PCG32 (`RNG rng;`, :372), so the directions and samples are the reference's.
// ---------------------------------------------------------------------------
namespace chi2 {
// Regularized lower incomplete gamma function (:44-113, after Cephes)
inline double RLGamma(double a, double x) {
    const double epsilon = 0.000000000000001, big = 4503599627370496.0, bigInv = 2.22044604925031308085e-16;
    if (a < 0 || x < 0) return 0.0;      // (the reference throws; never reached)
    if (x == 0) return 0;
    double ax = (a * std::log(x)) - x - std::lgamma(a);
    if (ax < -709.78271289338399) return a < x ? 1.0 : 0.0;
    if (x <= 1 || x <= a) {
        double r2 = a, c2 = 1, ans2 = 1;
        do { r2 = r2 + 1; c2 = c2 * x / r2; ans2 += c2; } while ((c2 / ans2) > epsilon);
        return std::exp(ax) * ans2 / a;
    }
    int c = 0;
    double y = 1 - a, z = x + y + 1, p3 = 1, q3 = x, p2 = x + 1, q2 = z * x, ans = p2 / q2, error;
    do {
        c++; y += 1; z += 2;
        double yc = y * c, p = (p2 * z) - (p3 * yc), q = (q2 * z) - (q3 * yc);
        if (q != 0) { double nextans = p / q; error = std::abs((ans - nextans) / nextans); ans = nextans; }
        else error = 1;
        p3 = p2; p2 = p; q3 = q2; q2 = q;
        if (std::abs(p) > big) { p3 *= bigInv; p2 *= bigInv; q3 *= bigInv; q2 *= bigInv; }
    } while (error > epsilon);
    return 1.0 - (std::exp(ax) * ans);
}
inline double Chi2CDF(double x, int dof) {      // :116-124
    if (dof < 1 || x < 0) return 0.0;
    if (dof == 2) return 1.0 - std::exp(-0.5 * x);
    return (Float)RLGamma(0.5 * dof, 0.5 * x);
}
template <typename F> Float SimpsonRec(const F &f, Float a, Float b, Float c, Float fa, Float fb, Float fc, Float I, Float eps, int depth) {   // :131-152
    Float d = 0.5f * (a + b), e = 0.5f * (b + c), fd = f(d), fe = f(e);
    Float h = c - a, I0 = (Float)(1.0 / 12.0) * h * (fa + 4 * fd + fb), I1 = (Float)(1.0 / 12.0) * h * (fb + 4 * fe + fc), Ip = I0 + I1;
    if (depth <= 0 || std::abs(Ip - I) < 15 * eps) return Ip + (Float)(1.0 / 15.0) * (Ip - I);
    return SimpsonRec(f, a, d, b, fa, fd, fb, I0, .5f * eps, depth - 1) + SimpsonRec(f, b, e, c, fb, fe, fc, I1, .5f * eps, depth - 1);
}
template <typename F> Float AdaptiveSimpson(const F &f, Float x0, Float x1, Float eps = 1e-6f, int depth = 6) {   // :127-159
    Float a = x0, b = 0.5f * (x0 + x1), c = x1;
    Float fa = f(a), fb = f(b), fc = f(c);
    Float I = (c - a) * (Float)(1.0 / 6.0) * (fa + 4 * fb + fc);
    return SimpsonRec(f, a, b, c, fa, fb, fc, I, eps, depth);
}
template <typename F> Float AdaptiveSimpson2D(const F &f, Float x0, Float y0, Float x1, Float y1, Float eps = 1e-6f, int depth = 6) {   // :162-173
    auto integrate = [&](Float y) { return AdaptiveSimpson([&](Float x) { return f(x, y); }, x0, x1, eps, depth); };
    return AdaptiveSimpson(integrate, y0, y1, eps, depth);
}
// :248-345
inline bool Chi2Test(const Float *frequencies, const Float *expFrequencies, int thetaRes, int phiRes, int sampleCount, Float minExpFrequency,
                     Float significanceLevel, int numTests, double *pvalOut) {
    struct Cell { Float expFrequency; size_t index; };
    std::vector<Cell> cells((size_t)thetaRes * phiRes);
    for (size_t i = 0; i < cells.size(); ++i) { cells[i].expFrequency = expFrequencies[i]; cells[i].index = i; }
    std::sort(cells.begin(), cells.end(), [](const Cell &a, const Cell &b) { return a.expFrequency < b.expFrequency; });
    Float pooledFrequencies = 0, pooledExpFrequencies = 0, chsq = 0;
    int pooledCells = 0, dof = 0;
    for (const Cell &c : cells) {
        if (expFrequencies[c.index] == 0) {
            if (frequencies[c.index] > sampleCount * 1e-5f) { *pvalOut = -1; return false; }
        } else if (expFrequencies[c.index] < minExpFrequency) {
            pooledFrequencies += frequencies[c.index]; pooledExpFrequencies += expFrequencies[c.index]; pooledCells++;
        } else if (pooledExpFrequencies > 0 && pooledExpFrequencies < minExpFrequency) {
            pooledFrequencies += frequencies[c.index]; pooledExpFrequencies += expFrequencies[c.index]; pooledCells++;
        } else {
            Float diff = frequencies[c.index] - expFrequencies[c.index];
            chsq += (diff * diff) / expFrequencies[c.index];
            ++dof;
        }
    }
    if (pooledExpFrequencies > 0 || pooledFrequencies > 0) {
        Float diff = pooledFrequencies - pooledExpFrequencies;
        chsq += (diff * diff) / pooledExpFrequencies;
        ++dof;
    }
    (void)pooledCells;
    dof -= 1;
    if (dof <= 0) { *pvalOut = -2; return false; }
    Float pval = 1 - (Float)Chi2CDF(chsq, dof);
    Float alpha = 1.0f - std::pow(1.0f - significanceLevel, 1.0f / numTests);
    *pvalOut = pval;
    return !(pval < alpha || !std::isfinite(pval));
}
}  // namespace chi2

// which: 0 Lambertian, 1 TR_VA_0p5, 2 TR_VA_0p3_0p15, 3 TR_VA_0p3 (FresnelBlend), 4 / 5 rough dielectric (below).  Returns the number of runs (of 5) whose null
// hypothesis was rejected; minPval: the smallest p-value met.
inline int SelfTestBSDFSampling(int which, double *minPval) {
    const int thetaRes = 10, phiRes = 20, sampleCount = 1000000, runs = 5;      // CHI2_* (:20-41)
    // the frame of the reference's disk hit (RotateX(-90) disk at y = 0 hit from above, :377-392): any orthonormal frame gives the
    // same local-space statistics; ns = +y with dpdu along the disk's tangent at (0.1, 0, 0)
    BSDF bsdf;
    bsdf.eta = 1; bsdf.ns = V3(0, 1, 0); bsdf.ng = V3(0, 1, 0); bsdf.ss = V3(0, 0, -1); bsdf.ts = Cross(bsdf.ns, bsdf.ss); bsdf.nBxDFs = 1;
    BxDF &b = bsdf.bxdfs[0];
    b = BxDF();
    if (which == 0) { b.kind = BXDF_LAMBERT; b.type = BSDF_REFLECTION | BSDF_DIFFUSE; b.R = Spec(1.f); }
    else if (which == 1 || which == 2) {
        b.kind = BXDF_MICROFACET; b.type = BSDF_REFLECTION | BSDF_GLOSSY; b.R = Spec(1.f);
        b.dist.alphax = RoughnessToAlpha(which == 1 ? 0.5f : 0.3f); b.dist.alphay = RoughnessToAlpha(which == 1 ? 0.5f : 0.15f);
    } else if (which == 3) {
        b.kind = BXDF_FRESNEL_BLEND; b.type = BSDF_REFLECTION | BSDF_GLOSSY; b.R = Spec(0.5f); b.S = Spec(0.5f);
        b.dist.alphax = RoughnessToAlpha(0.3f); b.dist.alphay = RoughnessToAlpha(0.3f);
    } else {
        // (not among the reference's cases: its chi-square METHOD applied to the rough-dielectric lobes it does not test.)  4: MicrofacetTransmission
        // alone, 5: the pair GlassMaterial builds — MicrofacetReflection(FresnelDielectric(1, 1.5)) + MicrofacetTransmission (materials/glass.cpp:66-93)
        bsdf.eta = 1.5f;
        if (which == 5) {
            b.kind = BXDF_MICROFACET; b.type = BSDF_REFLECTION | BSDF_GLOSSY; b.R = Spec(1.f);
            b.dist.alphax = RoughnessToAlpha(0.3f); b.dist.alphay = RoughnessToAlpha(0.15f); b.frDielectric = true; b.frEtaI = 1.f; b.frEtaT = 1.5f;
            bsdf.nBxDFs = 2;
        }
        BxDF &t = bsdf.bxdfs[which == 5 ? 1 : 0];
        t = BxDF();
        t.kind = BXDF_MICROFACET_TRANSMISSION; t.type = BSDF_TRANSMISSION | BSDF_GLOSSY; t.R = Spec(1.f);
        t.dist.alphax = RoughnessToAlpha(0.3f); t.dist.alphay = RoughnessToAlpha(which == 5 ? 0.15f : 0.3f); t.etaA = 1.f; t.etaB = 1.5f;
    }
    std::vector<Float> frequencies((size_t)thetaRes * phiRes), expFrequencies((size_t)thetaRes * phiRes);
    RNG rng;
    int rejected = 0;
    *minPval = 1;
    for (int k = 0; k < runs; ++k) {
        P2 sample; sample.x = rng.UniformFloat(); sample.y = rng.UniformFloat();
        V3 wo = bsdf.LocalToWorld(CosineSampleHemisphere(sample));
        // FrequencyTable (:176-201)
        std::fill(frequencies.begin(), frequencies.end(), (Float)0);
        const Float factorTheta = thetaRes / Pi, factorPhi = phiRes / (2 * Pi);
        for (int i = 0; i < sampleCount; ++i) {
            P2 u; u.x = rng.UniformFloat(); u.y = rng.UniformFloat();
            V3 wi; Float pdf; int flags = 0;
            Spec f = bsdf.Sample_f(wo, &wi, u, &pdf, BSDF_ALL, &flags);
            if (f.IsBlack() || (flags & BSDF_SPECULAR)) continue;
            V3 wiL = bsdf.WorldToLocal(wi);
            Float cx = std::acos(Clamp(wiL.z, -1, 1)) * factorTheta, cy = std::atan2(wiL.y, wiL.x) * factorPhi;
            if (cy < 0) cy += 2 * Pi * factorPhi;
            int thetaBin = std::min(std::max(0, (int)std::floor(cx)), thetaRes - 1);
            int phiBin = std::min(std::max(0, (int)std::floor(cy)), phiRes - 1);
            frequencies[(size_t)thetaBin * phiRes + phiBin] += 1;
        }
        // IntegrateFrequencyTable (:205-229)
        const Float cellTheta = Pi / thetaRes, cellPhi = (2 * Pi) / phiRes;
        for (int i = 0; i < thetaRes; ++i)
            for (int j = 0; j < phiRes; ++j)
                expFrequencies[(size_t)i * phiRes + j] = sampleCount * chi2::AdaptiveSimpson2D(
                    [&](Float theta, Float phi) -> Float {
                        Float cosTheta = std::cos(theta), sinTheta = std::sin(theta), cosPhi = std::cos(phi), sinPhi = std::sin(phi);
                        V3 wiL(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
                        return bsdf.Pdf(wo, bsdf.LocalToWorld(wiL), BSDF_ALL) * sinTheta;
                    }, i * cellTheta, j * cellPhi, (i + 1) * cellTheta, (j + 1) * cellPhi);
        double pval;
        if (!chi2::Chi2Test(frequencies.data(), expFrequencies.data(), thetaRes, phiRes, sampleCount, 5, 0.01f, runs, &pval)) ++rejected;
        if (pval < *minPval) *minPval = pval;
    }
    return rejected;
}

}  // namespace orc
