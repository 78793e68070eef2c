// hprt device side — ray/primitive tests and the ordered-stack BVH walk.
// Restates, operation for operation: Bounds3::IntersectP (core/geometry.h:1754-1780),
// Triangle::Intersect/IntersectP test part (shapes/triangle.cpp:193-292 / :431-530),
// Sphere::Intersect/IntersectP test part with EFloat bounds (shapes/sphere.cpp:49-104,
// core/efloat.h), BVHAccel::Intersect/IntersectP (accelerators/bvh.cpp:354-437).
#pragma once
#include "dev_scene.h"

namespace hprt {

struct DRay { vec3 o, d; float tMax; };

// ---- triangle -------------------------------------------------------------
// Ray-only part of the watertight test (shapes/triangle.cpp:207-221): the permutation
// and the shear coefficients depend on the ray alone, so they are computed once per ray
// (same operations, same values) instead of once per triangle test.
struct RayShear { bool k0, k1; float Sx, Sy, Sz; };     // k0: kz == 0, k1: kz == 1
// (with the ray's 1 / d at hand — the walk keeps it for the slab tests: Sz = 1.f / d[kz] IS invDir[kz], the same IEEE quotient: one division fewer)
__device__ __forceinline__ RayShear ray_shear(vec3 rd, vec3 invDir) {
    RayShear s;
    const int kz = max_dim(vabs(rd));
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    const float dx = rd.get(kx), dy = rd.get(ky), dz = rd.get(kz);
    s.k0 = kz == 0; s.k1 = kz == 1;
    s.Sx = -dx / dz; s.Sy = -dy / dz; s.Sz = invDir.get(kz);
    return s;
}
__device__ __forceinline__ RayShear ray_shear(vec3 rd) {
    RayShear s;
    const int kz = max_dim(vabs(rd));
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    const float dx = rd.get(kx), dy = rd.get(ky), dz = rd.get(kz);
    s.k0 = kz == 0; s.k1 = kz == 1;
    s.Sx = -dx / dz; s.Sy = -dy / dz; s.Sz = 1.f / dz;
    return s;
}
// Permute(v, kx, ky, kz) as plain selects.  kz == 0: (y,z,x); kz == 1: (z,x,y); kz == 2: (x,y,z)
__device__ __forceinline__ vec3 permute3(vec3 v, bool k0, bool k1) {
    const float a = k1 ? v.z : v.x, b = k1 ? v.x : v.y, c = k1 ? v.y : v.z;
    return vec3(k0 ? v.y : a, k0 ? v.z : b, k0 ? v.x : c);
}
__device__ __forceinline__ bool tri_test(vec3 p0, vec3 p1, vec3 p2, vec3 rayO, float rayTMax, const RayShear &sh, float *b0o,
                                         float *b1o, float *b2o, float *to) {
    vec3 p0t = p0 - rayO, p1t = p1 - rayO, p2t = p2 - rayO;
    p0t = permute3(p0t, sh.k0, sh.k1);
    p1t = permute3(p1t, sh.k0, sh.k1);
    p2t = permute3(p2t, sh.k0, sh.k1);
    const float Sx = sh.Sx, Sy = sh.Sy, Sz = sh.Sz;
    p0t.x += Sx * p0t.z; p0t.y += Sy * p0t.z;
    p1t.x += Sx * p1t.z; p1t.y += Sy * p1t.z;
    p2t.x += Sx * p2t.z; p2t.y += Sy * p2t.z;
    float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {   // double-precision fallback at edges
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
    float det = e0 + e1 + e2;
    if (det == 0) return false;
    p0t.z *= Sz; p1t.z *= Sz; p2t.z *= Sz;
    float tScaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if (det < 0 && (tScaled >= 0 || tScaled < rayTMax * det)) return false;
    else if (det > 0 && (tScaled <= 0 || tScaled > rayTMax * det)) return false;
    float invDet = 1 / det;
    float b0 = e0 * invDet, b1 = e1 * invDet, b2 = e2 * invDet;
    float t = tScaled * invDet;
    float maxZt = max_comp(vabs(vec3(p0t.z, p1t.z, p2t.z)));
    float deltaZ = gamma_n(3) * maxZt;
    float maxXt = max_comp(vabs(vec3(p0t.x, p1t.x, p2t.x)));
    float maxYt = max_comp(vabs(vec3(p0t.y, p1t.y, p2t.y)));
    float deltaX = gamma_n(5) * (maxXt + maxZt);
    float deltaY = gamma_n(5) * (maxYt + maxZt);
    float deltaE = 2 * (gamma_n(2) * maxXt * maxYt + deltaY * maxXt + deltaX * maxYt);
    float maxE = max_comp(vabs(vec3(e0, e1, e2)));
    float deltaT = 3 * (gamma_n(3) * maxE * maxZt + deltaE * maxZt + deltaZ * maxE) * fabsf(invDet);
    if (t <= deltaT) return false;
    *b0o = b0; *b1o = b1; *b2o = b2; *to = t;
    return true;
}

// ---- EFloat interval arithmetic (core/efloat.h, NDEBUG layout) --------------
struct efloat { float v, lo, hi; };
__device__ __forceinline__ efloat ef(float v) { efloat r; r.v = v; r.lo = v; r.hi = v; return r; }
__device__ __forceinline__ efloat ef(float v, float err) {
    efloat r; r.v = v;
    if (err == 0.f) { r.lo = v; r.hi = v; } else { r.lo = next_down(v - err); r.hi = next_up(v + err); }
    return r;
}
// The interval operations are calls of their own: sphere_test then needs 56 registers instead of 99, and its callers (a caller
// keeps its live values above its callee's registers) fit five waves per SIMD where that pays (kernels.hip).  The test itself is
// slower that way, but it runs for few rays: the pre-test below settles most.  (HPRT_EF_INLINE: the inlined form, for A/B builds.)
#ifdef HPRT_EF_INLINE
#define HPRT_EF_FN __device__ __forceinline__
#else
#define HPRT_EF_FN __device__ __noinline__
#endif
HPRT_EF_FN efloat ef_add(efloat a, efloat b) { efloat r; r.v = a.v + b.v; r.lo = next_down(a.lo + b.lo); r.hi = next_up(a.hi + b.hi); return r; }
HPRT_EF_FN efloat ef_sub(efloat a, efloat b) { efloat r; r.v = a.v - b.v; r.lo = next_down(a.lo - b.hi); r.hi = next_up(a.hi - b.lo); return r; }
HPRT_EF_FN efloat ef_mul(efloat a, efloat b) {
    efloat r; r.v = a.v * b.v;
    float p0 = a.lo * b.lo, p1 = a.hi * b.lo, p2 = a.lo * b.hi, p3 = a.hi * b.hi;
    r.lo = next_down(sel_min(sel_min(p0, p1), sel_min(p2, p3)));
    r.hi = next_up(sel_max(sel_max(p0, p1), sel_max(p2, p3)));
    return r;
}
HPRT_EF_FN efloat ef_div(efloat a, efloat b) {
    efloat r; r.v = a.v / b.v;
    if (b.lo < 0 && b.hi > 0) { r.lo = -HPRT_INF; r.hi = HPRT_INF; }
    else {
        float d0 = a.lo / b.lo, d1 = a.hi / b.lo, d2 = a.lo / b.hi, d3 = a.hi / b.hi;
        r.lo = next_down(sel_min(sel_min(d0, d1), sel_min(d2, d3)));
        r.hi = next_up(sel_max(sel_max(d0, d1), sel_max(d2, d3)));
    }
    return r;
}
// core/efloat.h:267-288
__device__ __forceinline__ bool ef_quadratic(efloat A, efloat B, efloat C, efloat *t0, efloat *t1) {
    double discrim = (double)B.v * (double)B.v - 4. * (double)A.v * (double)C.v;
    if (discrim < 0.) return false;
    double rootDiscrim = sqrt(discrim);
    efloat fr = ef((float)rootDiscrim, (float)((double)HPRT_MACHINE_EPS * rootDiscrim));
    efloat q;
    if (B.v < 0) q = ef_mul(ef(-.5f), ef_sub(B, fr));
    else q = ef_mul(ef(-.5f), ef_add(B, fr));
    *t0 = ef_div(q, A);
    *t1 = ef_div(C, q);
    if (t0->v > t1->v) { efloat tmp = *t0; *t0 = *t1; *t1 = tmp; }
    return true;
}

// "phi > phiMax" of the clipping test (shapes/sphere.cpp:93-97) without evaluating atan2 when
// the outcome is already certain.  For a (nearly) full sphere, phiMax > 6.283: with y >= 0 the
// angle is in [0, pi]; with y < 0 and x <= 0 it is in [pi, 3pi/2]; with y < 0 < x and
// -y/x > 1e-3 it is below 2pi - 9.9e-4.  None of these can exceed phiMax, so the comparison is
// false exactly as if phi had been computed.  Everything else takes the full evaluation.
__device__ __forceinline__ bool sphere_phi_exceeds(const DevSphere &s, vec3 pHit) {
    if (s.phiMax > 6.283f) {
        if (pHit.y >= 0.f) return false;
        if (pHit.x <= 0.f) return false;
        if (-pHit.y > 1e-3f * pHit.x) return false;
    }
    float phi = det_atan2f(pHit.y, pHit.x);
    if (phi < 0) phi += 2 * HPRT_PI;
    return phi > s.phiMax;
}

// The values (".v" lanes) of the quadric test's EFloat arithmetic up to its first rejection tests (shapes/sphere.cpp:55-75),
// in plain float.  An EFloat's bounds enclose its value (every operation keeps lo <= v <= hi), so t0.v > tMax implies the
// reference's t0.UpperBound() > tMax and t1.v <= 0 implies t1.LowerBound() <= 0; the discriminant test only reads values.
// false: the full test would return false; true: it has to run.  Most rays that reach the emitter's leaf end here: a shadow
// ray stops short of the sampled point (t0 ~ 1 > tMax = 1 - eps), most other rays miss the sphere or have it behind them.
__device__ __forceinline__ bool sphere_may_hit(const DevSphere &s, const DRay &r) {
    vec3 oErr;
    vec3 o = xf_point_err(s.w2o, r.o, &oErr);
    const vec3 d = xf_vector(s.w2o, r.d);
    const float len2 = length2(d);
    if (len2 > 0) { const float dt = dot(vabs(d), oErr) / len2; o = o + d * dt; }
    const float a = (d.x * d.x + d.y * d.y) + d.z * d.z;
    const float b = 2.f * ((d.x * o.x + d.y * o.y) + d.z * o.z);
    const float c = ((o.x * o.x + o.y * o.y) + o.z * o.z) - s.radius * s.radius;
    const double discrim = (double)b * (double)b - 4. * (double)a * (double)c;
    if (discrim < 0.) return false;
    const float fr = (float)sqrt(discrim);
    const float q = b < 0 ? -.5f * (b - fr) : -.5f * (b + fr);
    float t0 = q / a, t1 = c / q;
    if (t0 > t1) { const float tmp = t0; t0 = t1; t1 = tmp; }
    if (t0 > r.tMax || t1 <= 0) return false;
    return true;
}

// Quadric test shared by Sphere::Intersect and IntersectP (shapes/sphere.cpp:49-104 == :159-213).
__device__ __noinline__ bool sphere_test(const DevSphere &s, const DRay &r, DRay *rayObj, vec3 *pHitOut, float *phiOut,
                                         float *tOut) {
    vec3 oErr, dErr;
    // Transform::operator()(Ray, oError, dError), core/transform.h:349-360
    vec3 o = xf_point_err(s.w2o, r.o, &oErr);
    vec3 d = xf_vector_err(s.w2o, r.d, &dErr);
    float len2 = length2(d);
    if (len2 > 0) { float dt = dot(vabs(d), oErr) / len2; o = o + d * dt; }
    DRay ray; ray.o = o; ray.d = d; ray.tMax = r.tMax;
    efloat ox = ef(ray.o.x, oErr.x), oy = ef(ray.o.y, oErr.y), oz = ef(ray.o.z, oErr.z);
    efloat dx = ef(ray.d.x, dErr.x), dy = ef(ray.d.y, dErr.y), dz = ef(ray.d.z, dErr.z);
    efloat a = ef_add(ef_add(ef_mul(dx, dx), ef_mul(dy, dy)), ef_mul(dz, dz));
    efloat b = ef_mul(ef(2.f), ef_add(ef_add(ef_mul(dx, ox), ef_mul(dy, oy)), ef_mul(dz, oz)));
    efloat c = ef_sub(ef_add(ef_add(ef_mul(ox, ox), ef_mul(oy, oy)), ef_mul(oz, oz)), ef_mul(ef(s.radius), ef(s.radius)));
    efloat t0, t1;
    if (!ef_quadratic(a, b, c, &t0, &t1)) return false;
    if (t0.hi > ray.tMax || t1.lo <= 0) return false;
    efloat tHit = t0;
    if (tHit.lo <= 0) { tHit = t1; if (tHit.hi > ray.tMax) return false; }
    vec3 pHit = ray.o + ray.d * tHit.v;
    pHit = pHit * (s.radius / dist(pHit, vec3(0, 0, 0)));
    if (pHit.x == 0 && pHit.y == 0) pHit.x = 1e-5f * s.radius;
    if ((s.zMin > -s.radius && pHit.z < s.zMin) || (s.zMax < s.radius && pHit.z > s.zMax) || sphere_phi_exceeds(s, pHit)) {
        if (tHit.v == t1.v) return false;
        if (t1.hi > ray.tMax) return false;
        tHit = t1;
        pHit = ray.o + ray.d * tHit.v;
        pHit = pHit * (s.radius / dist(pHit, vec3(0, 0, 0)));
        if (pHit.x == 0 && pHit.y == 0) pHit.x = 1e-5f * s.radius;
        if ((s.zMin > -s.radius && pHit.z < s.zMin) || (s.zMax < s.radius && pHit.z > s.zMax) || sphere_phi_exceeds(s, pHit)) return false;
    }
    *rayObj = ray; *pHitOut = pHit; *phiOut = 0.f; *tOut = tHit.v;   // phi itself is not needed downstream (u is unused)
    return true;
}

// ---- BVH walk ---------------------------------------------------------------
// Per-lane traversal stack of {node reference, entry distance}: the first LDS_STACK
// entries live in LDS, laid out [entry][thread] so a wave's 64 lanes hit 64 consecutive
// 8-byte words; deeper entries spill to a private array (the reference reserves 64 entries,
// accelerators/bvh.cpp:362).  Only children whose slabs are hit are pushed, so the depth in
// use stays well below the tree depth.
#define HPRT_LDS_STACK 16
// the plain any-hit kernel of triangle-only scenes needs 66 registers: with a 10-entry LDS stack (20 KB per workgroup) seven
// workgroups fit a CU instead of five
#ifndef HPRT_LDS_STACK_ANY
#define HPRT_LDS_STACK_ANY 10
#endif
#define HPRT_STACK_TOTAL 64
#define HPRT_SPILL_STACK (HPRT_STACK_TOTAL - 7)      // deepest HBM part any kernel can need (the shortest LDS stack, instanced any-hit, keeps 7 entries)
// threads of the largest trace grid (256 CUs x 7 workgroups x 256 threads): stride of the deep-stack area, DevScene::deepStack
#define HPRT_DEEP_THREADS 458752u
#ifndef HPRT_TRACE_BLOCK
#define HPRT_TRACE_BLOCK 256
#endif

// The leaf-exact walk (k_walk4, dev_wide.h): LDS stack entries per lane — {ref, entry distance} pairs for closest-hit rays, bare
// references for any-hit rays — and the deepest stack a scene's wide tree may need (LDS + the deep-stack area)
#ifndef HPRT_WIDE_LDS_CLOSEST
#define HPRT_WIDE_LDS_CLOSEST 12
#endif
#ifndef HPRT_WIDE_LDS_ANY
#define HPRT_WIDE_LDS_ANY 20
#endif
#define HPRT_WIDE_STACK_MAX 60
static_assert(HPRT_WIDE_STACK_MAX - HPRT_WIDE_LDS_CLOSEST <= HPRT_SPILL_STACK && HPRT_WIDE_STACK_MAX - HPRT_WIDE_LDS_ANY <= HPRT_SPILL_STACK, "deep-stack area too small for the wide walk");

struct TraceCount { unsigned int fetched, entered, tri, sphere, leaf; };   // leaf: of the entered nodes, leaves

// `cur` of a lane: >= 0 interior pair, REF_NONE finished, REF_EXIT leaving an instance, otherwise ~(parked primitive index)
__device__ __forceinline__ bool is_parked(int cur) { return (uint32_t)cur > (uint32_t)REF_EXIT; }
// with object instances: parked primitives and the REF_EXIT sentinel are both handled by the primitive phase
__device__ __forceinline__ bool wants_prim_phase(int cur) { return (uint32_t)cur >= (uint32_t)REF_EXIT; }

}  // namespace hprt
