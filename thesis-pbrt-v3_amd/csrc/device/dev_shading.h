// hprt device side — everything PathIntegrator::Li evaluates at a path vertex:
// Halton sampler (samplers/halton.cpp, core/lowdiscrepancy.cpp), perspective camera
// (cameras/perspective.cpp:95-144), surface-interaction fill for triangles and
// spheres (shapes/triangle.cpp:294-425, shapes/sphere.cpp:106-157), matte/plastic
// BSDFs (core/reflection.cpp, core/microfacet.cpp), point/distant/sphere-area lights
// (lights/*.cpp, shapes/sphere.cpp:217-306) and ray spawning (core/interaction.h:64-78).
#pragma once
#include <cstddef>
#include "dev_intersect.h"

namespace hprt {

// ---------------------------------------------------------------------------
// Halton sampler
// ---------------------------------------------------------------------------
struct DevHalton {
    int32_t baseScale1;       // baseScales[1] (power of 3)
    int32_t baseExp0;         // baseExponents[0]
    int32_t sampleStride;
    int32_t samplePixelCenter;
};

// floor(a / base) for a < 2^32 via Lemire's 64-bit magic M = floor((2^64-1)/base)+1:
// q = (M * a) >> 64, exact for every 32-bit a.
__device__ __forceinline__ uint32_t div_magic32(uint32_t a, uint64_t M) {
    uint64_t t = (uint64_t)(uint32_t)M * a;
    uint64_t u = (M >> 32) * (uint64_t)a + (t >> 32);
    return (uint32_t)(u >> 32);
}
__device__ __forceinline__ uint64_t reverse_bits64(uint64_t n) {
    return ((uint64_t)__brev((uint32_t)n) << 32) | (uint64_t)__brev((uint32_t)(n >> 32));
}
// RadicalInverseSpecialized<base> / ScrambledRadicalInverseSpecialized<base>
// (core/lowdiscrepancy.cpp:389-424); perm == nullptr selects the unscrambled form.
// PermPtr is `const uint16_t *` (HBM table) or an LDS pointer (staged copy).
// LDS variant of the scrambled form: invBase = 1/(float)base and the infinite-tail term
// invBase*perm[0]/(1-invBase) depend on the dimension only, so they are evaluated once per
// block by halton_lds_load (same float operations, same values) instead of once per sample.
__device__ __forceinline__ float scrambled_radical_inverse_lds(uint32_t base, uint64_t M, uint32_t m32, uint64_t a, const uint16_t *perm, float invBase, float tail) {
    uint64_t reversedDigits = 0;
    float invBaseN = 1;
    if (a <= 0xffffffffull) {
        // 32-bit digit loop (every index below 2^32: e.g. up to 138,000 spp at the 128x128 Halton period).
        // q = mulhi(a, floor(2^32 / base)) is floor(a / base) or one less, fixed by one compare; the remainder is the
        // digit.  reversedDigits stays below 2^32 until the last digit is appended (it has one digit less than a
        // had, and a < 2^32), so each step is one 32 x 32 + 64 multiply-add.
        uint32_t a32 = (uint32_t)a;
        while (a32) {
            uint32_t q = __umulhi(a32, m32);
            uint32_t digit = a32 - q * base;
            if (digit >= base) { digit -= base; q += 1u; }
            reversedDigits = (uint64_t)(uint32_t)reversedDigits * (uint64_t)base + (uint64_t)perm[digit];
            invBaseN *= invBase;
            a32 = q;
        }
    } else {
        while (a) {
            uint64_t next;
            if (a <= 0xffffffffull) next = div_magic32((uint32_t)a, M);
            else next = a / base;
            uint32_t digit = (uint32_t)(a - next * base);
            reversedDigits = reversedDigits * base + (uint32_t)perm[digit];
            invBaseN *= invBase;
            a = next;
        }
    }
    // (float)uint64: a value below 2^32 converts identically through the 32-bit instruction
    const float rd = (reversedDigits >> 32) == 0ull ? (float)(uint32_t)reversedDigits : (float)reversedDigits;
    return sel_min(invBaseN * (rd + tail), HPRT_ONE_MINUS_EPS);
}
template <typename PermPtr>
__device__ __forceinline__ float radical_inverse_base(uint32_t base, uint64_t M, uint64_t a, PermPtr perm, bool scrambled) {
    const float invBase = 1.0f / (float)base;
    uint64_t reversedDigits = 0;
    float invBaseN = 1;
    while (a) {
        uint64_t next;
        if (a <= 0xffffffffull) next = div_magic32((uint32_t)a, M);
        else next = a / base;
        uint32_t digit = (uint32_t)(a - next * base);
        reversedDigits = reversedDigits * base + (scrambled ? (uint32_t)perm[digit] : digit);
        invBaseN *= invBase;
        a = next;
    }
    float v;
    if (scrambled) v = invBaseN * ((float)reversedDigits + invBase * (float)perm[0] / (1 - invBase));
    else v = (float)reversedDigits * invBaseN;
    return sel_min(v, HPRT_ONE_MINUS_EPS);
}
// Per-block LDS copy of the tables of the first HPRT_HALTON_LDS_DIMS dimensions (digit
// permutations: 8,893 uint16 = sum of the first 64 primes; bases; division magics).
// A maxdepth-5 path consumes at most 5 + 6*8 = 53 dimensions, so the scattered 2-byte
// permutation lookups of the digit loops are served from LDS instead of L2.
#define HPRT_HALTON_LDS_DIMS 64
#define HPRT_HALTON_LDS_PERMS 8893
struct alignas(16) HaltonLds {
    uint64_t magic[HPRT_HALTON_LDS_DIMS];
    uint32_t magic32[HPRT_HALTON_LDS_DIMS];      // floor(2^32 / prime)
    int32_t prime[HPRT_HALTON_LDS_DIMS];
    int32_t primeSum[HPRT_HALTON_LDS_DIMS];
    float invBase[HPRT_HALTON_LDS_DIMS];
    float tail[HPRT_HALTON_LDS_DIMS];
    uint16_t perm[HPRT_HALTON_LDS_PERMS + 3];
};
__device__ __forceinline__ void halton_lds_load(const DevScene &sc, HaltonLds *h) {
    for (int i = threadIdx.x; i < HPRT_HALTON_LDS_DIMS; i += blockDim.x) {
        h->magic[i] = sc.primeMagic[i]; h->magic32[i] = (uint32_t)(0x100000000ull / (uint64_t)(uint32_t)sc.primes[i]); h->prime[i] = sc.primes[i]; h->primeSum[i] = sc.primeSums[i];
        const float invBase = 1.0f / (float)sc.primes[i];
        h->invBase[i] = invBase;
        h->tail[i] = invBase * (float)sc.perms[sc.primeSums[i]] / (1 - invBase);   // invBase * perm[0] / (1 - invBase)
    }
    // the digit permutations as 16-byte words (1,112 of them: two or three loads per thread instead of seventeen 2-byte ones — this prologue
    // runs once per workgroup of 512 vertices, in front of everything)
    static_assert(((HPRT_HALTON_LDS_PERMS + 3) * sizeof(uint16_t)) % 16 == 0 && offsetof(HaltonLds, perm) % 16 == 0, "perm[] is copied as uint4");
    const uint4 *src = (const uint4 *)sc.perms;
    uint4 *dst = (uint4 *)h->perm;
    for (int i = threadIdx.x; i < (int)((HPRT_HALTON_LDS_PERMS + 3) * sizeof(uint16_t) / 16); i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}
// HaltonSampler::SampleDimension, samplers/halton.cpp:119-127
__device__ __forceinline__ float halton_dim(const DevScene &sc, const DevHalton &h, uint64_t index, int dim, const HaltonLds *lds = nullptr) {
    if (h.samplePixelCenter && (dim == 0 || dim == 1)) return 0.5f;
    if (dim == 0) return (float)((double)reverse_bits64(index >> h.baseExp0) * 0x1p-64);
    if (dim == 1) return radical_inverse_base<const uint16_t *>(3u, sc.primeMagic[1], index / (uint64_t)h.baseScale1, nullptr, false);
    if (lds && dim < HPRT_HALTON_LDS_DIMS)
        return scrambled_radical_inverse_lds((uint32_t)lds->prime[dim], lds->magic[dim], lds->magic32[dim], index, lds->perm + lds->primeSum[dim], lds->invBase[dim], lds->tail[dim]);
    return radical_inverse_base<const uint16_t *>((uint32_t)sc.primes[dim], sc.primeMagic[dim], index, sc.perms + sc.primeSums[dim], true);
}

// ---------------------------------------------------------------------------
// Camera
// ---------------------------------------------------------------------------
struct DevCamera { mat4 rasterToCamera, cameraToWorld; float lensRadius, focalDistance; vec3 dxCamera, dyCamera; };   // d*Camera: cameras/perspective.cpp:55-58

// core/sampling.cpp:113-130
__device__ __forceinline__ void concentric_disk(float ux, float uy, float *dx, float *dy) {
    float ox = 2.f * ux - 1, oy = 2.f * uy - 1;
    if (ox == 0 && oy == 0) { *dx = 0; *dy = 0; return; }
    float theta, r;
    if (fabsf(ox) > fabsf(oy)) { r = ox; theta = HPRT_PI_OVER_4 * (oy / ox); }
    else { r = oy; theta = HPRT_PI_OVER_2 - HPRT_PI_OVER_4 * (ox / oy); }
    float sn, cs; det_sincosf(theta, &sn, &cs);
    *dx = r * cs; *dy = r * sn;
}
// PerspectiveCamera::GenerateRayDifferential main ray + CameraToWorld(ray) (transform.h:245-259)
__device__ __forceinline__ void camera_ray(const DevCamera &cam, float fx, float fy, float lu, float lv, DRay *out) {
    vec3 pCamera = xf_point(cam.rasterToCamera, vec3(fx, fy, 0));
    vec3 ro(0, 0, 0), rd = normalize(vec3(pCamera.x, pCamera.y, pCamera.z));
    if (cam.lensRadius > 0) {
        float cx, cy; concentric_disk(lu, lv, &cx, &cy);
        float lx = cam.lensRadius * cx, ly = cam.lensRadius * cy;
        float ft = cam.focalDistance / rd.z;
        vec3 pFocus = ro + rd * ft;
        ro = vec3(lx, ly, 0);
        rd = normalize(pFocus - ro);
    }
    vec3 oErr;
    vec3 o = xf_point_err(cam.cameraToWorld, ro, &oErr);
    vec3 d = xf_vector(cam.cameraToWorld, rd);
    float len2 = length2(d), tMax = HPRT_INF;
    if (len2 > 0) { float dt = dot(vabs(d), oErr) / len2; o = o + d * dt; tMax -= dt; }
    out->o = o; out->d = d; out->tMax = tMax;
}

// The offset rays of GenerateRayDifferential (cameras/perspective.cpp:117-143), in world space
// (Transform::operator()(const RayDifferential &), core/transform.h:266-275); image textures only.
struct DevRayDiff { vec3 rxO, ryO, rxD, ryD; };
__device__ __forceinline__ void camera_ray_diff(const DevCamera &cam, float fx, float fy, float lu, float lv, DevRayDiff *rd) {
    const vec3 pCamera = xf_point(cam.rasterToCamera, vec3(fx, fy, 0));
    vec3 rxO(0, 0, 0), ryO(0, 0, 0), rxD, ryD;
    if (cam.lensRadius > 0) {
        float cx, cy; concentric_disk(lu, lv, &cx, &cy);
        const float lx = cam.lensRadius * cx, ly = cam.lensRadius * cy;
        const vec3 dx = normalize(pCamera + cam.dxCamera);
        float ft = cam.focalDistance / dx.z;
        vec3 pFocus = vec3(0, 0, 0) + (ft * dx);
        rxO = vec3(lx, ly, 0);
        rxD = normalize(pFocus - rxO);
        const vec3 dy = normalize(pCamera + cam.dyCamera);
        ft = cam.focalDistance / dy.z;
        pFocus = vec3(0, 0, 0) + (ft * dy);
        ryO = vec3(lx, ly, 0);
        ryD = normalize(pFocus - ryO);
    } else {
        rxD = normalize(pCamera + cam.dxCamera);
        ryD = normalize(pCamera + cam.dyCamera);
    }
    rd->rxO = xf_point(cam.cameraToWorld, rxO); rd->ryO = xf_point(cam.cameraToWorld, ryO);
    rd->rxD = xf_vector(cam.cameraToWorld, rxD); rd->ryD = xf_vector(cam.cameraToWorld, ryD);
}

// ---------------------------------------------------------------------------
// Surface interaction
// ---------------------------------------------------------------------------
struct DevSI { vec3 p, pErr, wo, n, ns, sdpdu; int32_t shape; };   // ns = shading.n, sdpdu = shading.dpdu
// what image textures additionally need of the interaction: (u,v) and the parametric derivatives
struct DevTexGeom { vec3 dpdu, dpdv; float u, v; };

// SurfaceInteraction::SetShadingGeometry with orientationIsAuthoritative (core/interaction.cpp:72-91)
__device__ __forceinline__ void set_shading(DevSI *si, vec3 dpdus, vec3 dpdvs, bool flip) {
    si->ns = normalize(cross(dpdus, dpdvs));
    if (flip) si->ns = -si->ns;
    si->n = face_forward(si->n, si->ns);
    si->sdpdu = dpdus;
}
// Triangle::Intersect fill part, shapes/triangle.cpp:294-425.  The degenerate
// (bogus) case was rejected during traversal via TAG_BOGUS.
__device__ __forceinline__ void fill_triangle(const DevScene &sc, uint32_t prim, float b0, float b1, float b2, vec3 rayD, DevSI *si,
                                              DevTexGeom *tg = nullptr) {
    const float4 v0 = sc.tris[3 * prim], v1 = sc.tris[3 * prim + 1], v2 = sc.tris[3 * prim + 2];
    const vec3 p0(v0.x, v0.y, v0.z), p1(v1.x, v1.y, v1.z), p2(v2.x, v2.y, v2.z);
    const int shapeId = (int)__float_as_uint(v1.w);
    const DevShape sh = sc.shapes[shapeId];
    const bool flip = (sh.flags & SHAPE_FLIP) != 0;
    uint32_t i0 = 0u, i1 = 0u, i2 = 0u;      // vertex ids: only the rare uv / tangent attributes are reached through them
    if (sh.flags & (SHAPE_HAS_UV | SHAPE_HAS_S)) { i0 = sc.primVtx[3 * prim]; i1 = sc.primVtx[3 * prim + 1]; i2 = sc.primVtx[3 * prim + 2]; }
    float uv0x = 0, uv0y = 0, uv1x = 1, uv1y = 0, uv2x = 1, uv2y = 1;   // triangle.h:114-118
    if (sh.flags & SHAPE_HAS_UV) {
        uv0x = sc.vUV[2 * i0]; uv0y = sc.vUV[2 * i0 + 1]; uv1x = sc.vUV[2 * i1]; uv1y = sc.vUV[2 * i1 + 1];
        uv2x = sc.vUV[2 * i2]; uv2y = sc.vUV[2 * i2 + 1];
    }
    float duv02x = uv0x - uv2x, duv02y = uv0y - uv2y, duv12x = uv1x - uv2x, duv12y = uv1y - uv2y;
    vec3 dp02 = p0 - p2, dp12 = p1 - p2;
    float determinant = duv02x * duv12y - duv02y * duv12x;
    bool degenerateUV = fabsf(determinant) < 1e-8;     // float |det| compared against the double literal
    vec3 dpdu, dpdv;
    if (!degenerateUV) {
        float invdet = 1 / determinant;
        dpdu = (duv12y * dp02 - duv02y * dp12) * invdet;
        dpdv = (-duv12x * dp02 + duv02x * dp12) * invdet;
    }
    if (degenerateUV || length2(cross(dpdu, dpdv)) == 0) {
        vec3 ng = cross(p2 - p0, p1 - p0);
        coordinate_system(normalize(ng), &dpdu, &dpdv);
    }
    float xAbs = (fabsf(b0 * p0.x) + fabsf(b1 * p1.x) + fabsf(b2 * p2.x));
    float yAbs = (fabsf(b0 * p0.y) + fabsf(b1 * p1.y) + fabsf(b2 * p2.y));
    float zAbs = (fabsf(b0 * p0.z) + fabsf(b1 * p1.z) + fabsf(b2 * p2.z));
    si->pErr = gamma_n(7) * vec3(xAbs, yAbs, zAbs);
    si->p = b0 * p0 + b1 * p1 + b2 * p2;
    si->wo = normalize(-rayD);
    si->shape = shapeId;
    si->n = normalize(cross(dp02, dp12));
    si->ns = si->n;
    si->sdpdu = dpdu;
    if (tg) {      // uvHit = b0 * uv[0] + b1 * uv[1] + b2 * uv[2] (shapes/triangle.cpp:328)
        tg->dpdu = dpdu; tg->dpdv = dpdv;
        tg->u = b0 * uv0x + b1 * uv1x + b2 * uv2x; tg->v = b0 * uv0y + b1 * uv1y + b2 * uv2y;
    }
    if (sh.flags & (SHAPE_HAS_N | SHAPE_HAS_S)) {
        vec3 ns;
        if (sh.flags & SHAPE_HAS_N) {
            const float4 m0 = sc.primN[3 * prim], m1 = sc.primN[3 * prim + 1], m2 = sc.primN[3 * prim + 2];
            const vec3 n0(m0.x, m0.y, m0.z), n1(m1.x, m1.y, m1.z), n2(m2.x, m2.y, m2.z);
            ns = (b0 * n0 + b1 * n1 + b2 * n2);
            if (length2(ns) > 0) ns = normalize(ns); else ns = si->n;
        } else ns = si->n;
        vec3 ss;
        if (sh.flags & SHAPE_HAS_S) {
            vec3 s0(sc.vS[3 * i0], sc.vS[3 * i0 + 1], sc.vS[3 * i0 + 2]), s1(sc.vS[3 * i1], sc.vS[3 * i1 + 1], sc.vS[3 * i1 + 2]),
                s2(sc.vS[3 * i2], sc.vS[3 * i2 + 1], sc.vS[3 * i2 + 2]);
            ss = (b0 * s0 + b1 * s1 + b2 * s2);
            if (length2(ss) > 0) ss = normalize(ss); else ss = normalize(dpdu);
        } else ss = normalize(dpdu);
        vec3 ts = cross(ss, ns);
        if (length2(ts) > 0.f) { ts = normalize(ts); ss = cross(ts, ns); }
        else coordinate_system(ns, &ss, &ts);
        set_shading(si, ss, ts, flip);
    }
    if (sh.flags & SHAPE_HAS_N) si->n = face_forward(si->n, si->ns);
    else if (flip) { si->n = -si->n; si->ns = si->n; }
}
// Sphere::Intersect fill part + (*ObjectToWorld)(SurfaceInteraction)
// (shapes/sphere.cpp:106-157, core/transform.cpp:262-297).  Returns false if the
// quadric test fails (cannot happen for a primitive the traversal reported).
__device__ __noinline__ bool fill_sphere(const DevScene &sc, int shapeId, const DRay &r, DevSI *si, float *tOut, DevTexGeom *tg = nullptr) {
    const DevShape sh = sc.shapes[shapeId];
    const DevSphere &s = sc.spheres[sh.sphere];
    DRay ray; vec3 pHit; float phi, t;
    if (!sphere_test(s, r, &ray, &pHit, &phi, &t)) return false;
    float theta = det_acosf(clampf(pHit.z / s.radius, -1, 1));
    float zRadius = sqrtf(pHit.x * pHit.x + pHit.y * pHit.y);
    float invZRadius = 1 / zRadius;
    float cosPhi = pHit.x * invZRadius, sinPhi = pHit.y * invZRadius;
    vec3 dpdu(-s.phiMax * pHit.y, s.phiMax * pHit.x, 0);
    vec3 dpdv = (s.thetaMax - s.thetaMin) * vec3(pHit.z * cosPhi, pHit.z * sinPhi, -s.radius * det_sinf(theta));
    vec3 pErrObj = gamma_n(5) * vabs(pHit);
    // SurfaceInteraction ctor (core/interaction.cpp:43-70)
    const bool flip = (sh.flags & SHAPE_FLIP) != 0;
    vec3 nObj = normalize(cross(dpdu, dpdv));
    vec3 nsObj = nObj;
    if (flip) { nObj = -nObj; nsObj = -nsObj; }
    vec3 woObj = normalize(-ray.d);
    si->p = xf_point_err_in(s.o2w, pHit, pErrObj, &si->pErr);
    si->n = normalize(xf_normal(s.w2o, nObj));
    si->wo = normalize(xf_vector(s.o2w, woObj));
    si->ns = normalize(xf_normal(s.w2o, nsObj));
    si->sdpdu = xf_vector(s.o2w, dpdu);
    si->ns = face_forward(si->ns, si->n);
    si->shape = shapeId;
    if (tg) {      // u = phi / phiMax, v = (theta - thetaMin) / (thetaMax - thetaMin) (shapes/sphere.cpp:108-110)
        phi = det_atan2f(pHit.y, pHit.x);      // sphere_test skips phi where the clipping test does not need it
        if (phi < 0) phi += 2 * HPRT_PI;
        tg->u = phi / s.phiMax; tg->v = (theta - s.thetaMin) / (s.thetaMax - s.thetaMin);
        tg->dpdu = xf_vector(s.o2w, dpdu); tg->dpdv = xf_vector(s.o2w, dpdv);
    }
    *tOut = t;
    return true;
}

// ---------------------------------------------------------------------------
// BSDF: up to two lobes (Lambertian, Trowbridge-Reitz microfacet with
// FresnelDielectric(1.5, 1)), as MatteMaterial / PlasticMaterial build them
// (materials/matte.cpp:45-62, materials/plastic.cpp:45-70).
// ---------------------------------------------------------------------------
enum : int { BX_REFLECTION = 1, BX_TRANSMISSION = 2, BX_DIFFUSE = 4, BX_GLOSSY = 8, BX_SPECULAR = 16, BX_ALL = 31 };

// No arrays here on purpose: a run-time-indexed member array would live in scratch memory.
struct DevBsdf {
    vec3 ns, ng, ss, ts;
    rgb Rd, Rs;            // Lambertian reflectance (lobe 0 when present), microfacet reflectance (the following lobe)
    float alpha;           // Trowbridge-Reitz alpha of the microfacet lobe
    bool hasD, hasS;
    bool hasR;             // a SpecularReflection lobe with FresnelNoOp (mirror): the BSDF's only lobe when present
    bool hasT;             // kind 6 (rough glass): a MicrofacetTransmission lobe (T in Rd) behind the MicrofacetReflection lobe (hasS: R in Rs)
    rgb Rr;
    bool oren; float orenA, orenB;      // the diffuse lobe is OrenNayar, not LambertianReflection
    float alphaY;          // Trowbridge-Reitz alpha along v (== alpha for plastic)
    float eta;             // BSDF::eta (1 unless glass)
    float frI, frT;        // FresnelDielectric(frI, frT) of the microfacet lobe: (1.5, 1) for plastic, (1, e) for uber
    rgb op;                    // kind 5: the clamped opacity at this vertex (a constant or its image texture, materials/uber.cpp:53)
    const DevMaterial *uber;   // kind 5 (UberMaterial): the material, for the specular lobes only Sample_f(BSDF_ALL) can pick (kr, kt, 1 - opacity)
    int kind;              // 0: the lobes above; 4 (with hasR): ONE FresnelSpecular lobe (glass: Rr = R, Rd = T, eta); 2: ONE FresnelBlend lobe (substrate: Rd, Rs, alpha, alphaY); 3: ONE conductor
                           // microfacet lobe (metal: Rd = eta, Rs = k, R = 1).  Kinds 2 and 3 are flagged hasS (a glossy reflection lobe).  6: rough glass — MicrofacetReflection
                           // (hasS: Rs = R, FresnelDielectric(1, eta)) and / or MicrofacetTransmission (hasT: Rd = T) over one Trowbridge-Reitz distribution (materials/glass.cpp:61-93)
};
__device__ __forceinline__ float cos_theta(vec3 w) { return w.z; }
__device__ __forceinline__ float cos2_theta(vec3 w) { return w.z * w.z; }
__device__ __forceinline__ float abs_cos_theta(vec3 w) { return fabsf(w.z); }
__device__ __forceinline__ float sin2_theta(vec3 w) { return sel_max(0.f, 1.0f - cos2_theta(w)); }
__device__ __forceinline__ float sin_theta(vec3 w) { return sqrtf(sin2_theta(w)); }
__device__ __forceinline__ float tan_theta(vec3 w) { return sin_theta(w) / cos_theta(w); }
__device__ __forceinline__ float tan2_theta(vec3 w) { return sin2_theta(w) / cos2_theta(w); }
__device__ __forceinline__ float cos_phi(vec3 w) { float s = sin_theta(w); return (s == 0) ? 1 : clampf(w.x / s, -1, 1); }
__device__ __forceinline__ float sin_phi(vec3 w) { float s = sin_theta(w); return (s == 0) ? 0 : clampf(w.y / s, -1, 1); }
__device__ __forceinline__ float cos2_phi(vec3 w) { return cos_phi(w) * cos_phi(w); }
__device__ __forceinline__ float sin2_phi(vec3 w) { return sin_phi(w) * sin_phi(w); }
__device__ __forceinline__ bool same_hemisphere(vec3 w, vec3 wp) { return w.z * wp.z > 0; }

// core/reflection.cpp:47-68
__device__ __forceinline__ float fr_dielectric(float cosThetaI, float etaI, float etaT) {
    cosThetaI = clampf(cosThetaI, -1, 1);
    bool entering = cosThetaI > 0.f;
    if (!entering) { float tmp = etaI; etaI = etaT; etaT = tmp; cosThetaI = fabsf(cosThetaI); }
    float sinThetaI = sqrtf(sel_max(0.f, 1 - cosThetaI * cosThetaI));
    float sinThetaT = etaI / etaT * sinThetaI;
    if (sinThetaT >= 1) return 1;
    float cosThetaT = sqrtf(sel_max(0.f, 1 - sinThetaT * sinThetaT));
    float Rparl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
    float Rperp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
    return (Rparl * Rparl + Rperp * Rperp) / 2;
}
// TrowbridgeReitzDistribution (core/microfacet.cpp:163-193, 238-344)
__device__ __forceinline__ float tr_D(float ax, float ay, vec3 wh) {
    float tan2Theta = tan2_theta(wh);
    if (is_inf(tan2Theta)) return 0.;
    const float cos4Theta = cos2_theta(wh) * cos2_theta(wh);
    float e = (cos2_phi(wh) / (ax * ax) + sin2_phi(wh) / (ay * ay)) * tan2Theta;
    return 1 / (HPRT_PI * ax * ay * cos4Theta * (1 + e) * (1 + e));
}
__device__ __forceinline__ float tr_lambda(float ax, float ay, vec3 w) {
    float absTanTheta = fabsf(tan_theta(w));
    if (is_inf(absTanTheta)) return 0.;
    float alpha = sqrtf(cos2_phi(w) * ax * ax + sin2_phi(w) * ay * ay);
    float alpha2Tan2Theta = (alpha * absTanTheta) * (alpha * absTanTheta);
    return (-1 + sqrtf(1.f + alpha2Tan2Theta)) / 2;
}
__device__ __forceinline__ float tr_G1(float ax, float ay, vec3 w) { return 1 / (1 + tr_lambda(ax, ay, w)); }
__device__ __forceinline__ float tr_G(float ax, float ay, vec3 wo, vec3 wi) { return 1 / (1 + tr_lambda(ax, ay, wo) + tr_lambda(ax, ay, wi)); }
__device__ __forceinline__ float tr_pdf(float ax, float ay, vec3 wo, vec3 wh) { return tr_D(ax, ay, wh) * tr_G1(ax, ay, wo) * absdot(wo, wh) / abs_cos_theta(wo); }
__device__ __forceinline__ void tr_sample11(float cosTheta, float U1, float U2, float *slope_x, float *slope_y) {
    if ((double)cosTheta > .9999) {
        float r = sqrtf(U1 / (1 - U1));
        float phi = (float)(6.28318530718 * (double)U2);
        double sn, cs; det_sincos_glibc_d((double)phi, &sn, &cs);      // ::sin / ::cos of glibc, bit for bit
        *slope_x = (float)((double)r * cs);   // float * ::cos(double) -> double -> float
        *slope_y = (float)((double)r * sn);
        return;
    }
    float sinTheta = sqrtf(sel_max(0.f, 1.0f - cosTheta * cosTheta));
    float tanTheta = sinTheta / cosTheta;
    float a = 1 / tanTheta;
    float G1 = 2 / (1 + sqrtf(1.f + 1.f / (a * a)));
    float A = 2 * U1 / G1 - 1;
    float tmp = 1.f / (A * A - 1.f);
    if ((double)tmp > 1e10) tmp = (float)1e10;
    float B = tanTheta;
    float D = sqrtf(sel_max(B * B * tmp * tmp - (A * A - B * B) * tmp, 0.f));
    float slope_x_1 = B * tmp - D;
    float slope_x_2 = B * tmp + D;
    *slope_x = (A < 0 || slope_x_2 > 1.f / tanTheta) ? slope_x_1 : slope_x_2;
    float S;
    if (U2 > 0.5f) { S = 1.f; U2 = 2.f * (U2 - .5f); }
    else { S = -1.f; U2 = 2.f * (.5f - U2); }
    float z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) /
              (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
    *slope_y = S * z * sqrtf(1.f + *slope_x * *slope_x);
}
__device__ __forceinline__ vec3 tr_sample_wh(float ax, float ay, vec3 wo, float u0, float u1) {
    bool flip = wo.z < 0;
    vec3 wi = flip ? -wo : wo;
    vec3 wiS = normalize(vec3(ax * wi.x, ay * wi.y, wi.z));
    float sx, sy;
    tr_sample11(cos_theta(wiS), u0, u1, &sx, &sy);
    float tmp = cos_phi(wiS) * sx - sin_phi(wiS) * sy;
    sy = sin_phi(wiS) * sx + cos_phi(wiS) * sy;
    sx = tmp;
    sx = ax * sx; sy = ay * sy;
    vec3 wh = normalize(vec3(-sx, -sy, 1.f));
    if (flip) wh = -wh;
    return wh;
}

// LambertianReflection (core/reflection.cpp:178-180) with BxDF::Sample_f / BxDF::Pdf (:378-389)
// (or OrenNayar::f, core/reflection.cpp:197-219, when the matte material has sigma != 0)
__device__ __forceinline__ rgb lambert_f(const DevBsdf &b, vec3 wo, vec3 wi) {
    if (!b.oren) return b.Rd * HPRT_INV_PI;
    const float sinThetaI = sin_theta(wi), sinThetaO = sin_theta(wo);
    float maxCos = 0;
    if ((double)sinThetaI > 1e-4 && (double)sinThetaO > 1e-4) {
        const float sinPhiI = sin_phi(wi), cosPhiI = cos_phi(wi), sinPhiO = sin_phi(wo), cosPhiO = cos_phi(wo);
        const float dCos = cosPhiI * cosPhiO + sinPhiI * sinPhiO;
        maxCos = sel_max(0.f, dCos);
    }
    float sinAlpha, tanBeta;
    if (abs_cos_theta(wi) > abs_cos_theta(wo)) { sinAlpha = sinThetaO; tanBeta = sinThetaI / abs_cos_theta(wi); }
    else { sinAlpha = sinThetaI; tanBeta = sinThetaO / abs_cos_theta(wo); }
    return b.Rd * HPRT_INV_PI * (b.orenA + b.orenB * maxCos * sinAlpha * tanBeta);
}
__device__ __forceinline__ float lambert_pdf(vec3 wo, vec3 wi) { return same_hemisphere(wo, wi) ? abs_cos_theta(wi) * HPRT_INV_PI : 0; }
__device__ __forceinline__ rgb rgb_sqrt(rgb a) { return rgb(sqrtf(a.r), sqrtf(a.g), sqrtf(a.b)); }
__device__ __forceinline__ rgb rgb_sub(rgb a, rgb b) { return rgb(a.r - b.r, a.g - b.g, a.b - b.b); }
__device__ __forceinline__ rgb rgb_div(rgb a, rgb b) { return rgb(a.r / b.r, a.g / b.g, a.b / b.b); }
// FrConductor(cosThetaI, etai = 1, etat = eta, k), core/reflection.cpp:70-95
__device__ __forceinline__ rgb fr_conductor(float cosThetaI, rgb etat, rgb k) {
    cosThetaI = clampf(cosThetaI, -1, 1);
    const rgb etai(1.f);
    const rgb eta = rgb_div(etat, etai), etak = rgb_div(k, etai);
    const float cosThetaI2 = cosThetaI * cosThetaI;
    const float sinThetaI2 = (float)(1. - (double)cosThetaI2);
    const rgb eta2 = eta * eta, etak2 = etak * etak;
    const rgb t0 = rgb_sub(rgb_sub(eta2, etak2), rgb(sinThetaI2));
    const rgb a2plusb2 = rgb_sqrt(t0 * t0 + (eta2 * 4.f) * etak2);
    const rgb t1 = a2plusb2 + rgb(cosThetaI2);
    const rgb a = rgb_sqrt((a2plusb2 + t0) * 0.5f);
    const rgb t2 = a * ((float)2 * cosThetaI);
    const rgb Rs = rgb_div(rgb_sub(t1, t2), t1 + t2);
    const rgb t3 = a2plusb2 * cosThetaI2 + rgb(sinThetaI2 * sinThetaI2);
    const rgb t4 = t2 * sinThetaI2;
    const rgb Rp = rgb_div(Rs * rgb_sub(t3, t4), t3 + t4);
    return (Rp + Rs) * 0.5f;
}
// MicrofacetReflection::f / Pdf (core/reflection.cpp:226-236, 416-420), FresnelDielectric(1.5, 1)
__device__ __forceinline__ rgb mf_f(const DevBsdf &b, vec3 wo, vec3 wi) {
    float cosThetaO = abs_cos_theta(wo), cosThetaI = abs_cos_theta(wi);
    vec3 wh = wi + wo;
    if (cosThetaI == 0 || cosThetaO == 0) return rgb(0.f);
    if (wh.x == 0 && wh.y == 0 && wh.z == 0) return rgb(0.f);
    wh = normalize(wh);
    // FresnelDielectric(1.5, 1) (plastic), or FresnelConductor(1, eta, k) with R = 1 (metal: Rd = eta, Rs = k)
    if (b.kind == 3) return rgb(1.f) * tr_D(b.alpha, b.alphaY, wh) * tr_G(b.alpha, b.alphaY, wo, wi) * fr_conductor(fabsf(dot(wi, wh)), b.Rd, b.Rs) / (4 * cosThetaI * cosThetaO);
    rgb F(fr_dielectric(dot(wi, wh), b.frI, b.frT));
    return b.Rs * tr_D(b.alpha, b.alphaY, wh) * tr_G(b.alpha, b.alphaY, wo, wi) * F / (4 * cosThetaI * cosThetaO);
}
__device__ __forceinline__ float mf_pdf(const DevBsdf &b, vec3 wo, vec3 wi) {
    if (!same_hemisphere(wo, wi)) return 0;
    vec3 wh = normalize(wo + wi);
    return tr_pdf(b.alpha, b.alphaY, wo, wh) / (4 * dot(wo, wh));
}

// FresnelBlend (core/reflection.cpp:285-298, 450-475): Rd, Rs, alpha, alphaY
__device__ __forceinline__ float pow5f(float v) { return (v * v) * (v * v) * v; }
__device__ __forceinline__ rgb blend_f(const DevBsdf &b, vec3 wo, vec3 wi) {
    const rgb diffuse = b.Rd * (28.f / (23.f * HPRT_PI)) * rgb_sub(rgb(1.f), b.Rs) * (1 - pow5f(1 - .5f * abs_cos_theta(wi))) * (1 - pow5f(1 - .5f * abs_cos_theta(wo)));
    vec3 wh = wi + wo;
    if (wh.x == 0 && wh.y == 0 && wh.z == 0) return rgb(0.f);
    wh = normalize(wh);
    const rgb schlick = b.Rs + rgb_sub(rgb(1.f), b.Rs) * pow5f(1 - dot(wi, wh));
    const rgb specular = schlick * (tr_D(b.alpha, b.alphaY, wh) / (4 * absdot(wi, wh) * sel_max(abs_cos_theta(wi), abs_cos_theta(wo))));
    return diffuse + specular;
}
__device__ __forceinline__ float blend_pdf(const DevBsdf &b, vec3 wo, vec3 wi) {
    if (!same_hemisphere(wo, wi)) return 0;
    const vec3 wh = normalize(wo + wi);
    const float pdf_wh = tr_pdf(b.alpha, b.alphaY, wo, wh);
    return .5f * (abs_cos_theta(wi) * HPRT_INV_PI + pdf_wh / (4 * dot(wo, wh)));
}
// MicrofacetTransmission (core/reflection.cpp:244-266, 425-447) with etaA = 1, etaB = b.eta, T = b.Rd, TransportMode::Radiance
__device__ __forceinline__ rgb mt_f(const DevBsdf &b, vec3 wo, vec3 wi) {
    if (same_hemisphere(wo, wi)) return rgb(0.f);      // transmission only
    const float cosThetaO = cos_theta(wo), cosThetaI = cos_theta(wi);
    if (cosThetaI == 0 || cosThetaO == 0) return rgb(0.f);
    const float eta = cos_theta(wo) > 0 ? (b.eta / 1.f) : (1.f / b.eta);
    vec3 wh = normalize(wo + wi * eta);
    if (wh.z < 0) wh = -wh;
    const rgb F(fr_dielectric(dot(wo, wh), 1.f, b.eta));
    const float sqrtDenom = dot(wo, wh) + eta * dot(wi, wh);
    const float factor = 1 / eta;
    return rgb_sub(rgb(1.f), F) * b.Rd *
           fabsf(tr_D(b.alpha, b.alphaY, wh) * tr_G(b.alpha, b.alphaY, wo, wi) * eta * eta * absdot(wi, wh) * absdot(wo, wh) * factor * factor /
                 (cosThetaI * cosThetaO * sqrtDenom * sqrtDenom));
}
__device__ __forceinline__ float mt_pdf(const DevBsdf &b, vec3 wo, vec3 wi) {
    if (same_hemisphere(wo, wi)) return 0;
    const float eta = cos_theta(wo) > 0 ? (b.eta / 1.f) : (1.f / b.eta);
    const vec3 wh = normalize(wo + wi * eta);
    const float sqrtDenom = dot(wo, wh) + eta * dot(wi, wh);
    const float dwh_dwi = fabsf((eta * eta * dot(wi, wh)) / (sqrtDenom * sqrtDenom));
    return tr_pdf(b.alpha, b.alphaY, wo, wh) * dwh_dwi;
}
// materials/matte.cpp:45-62, materials/plastic.cpp:45-70: lobes are added in the order
// diffuse, specular; a black reflectance adds no lobe.
// ---- image textures: SurfaceInteraction::ComputeDifferentials (core/interaction.cpp:103-149, the (u,v) part),
// UVMapping2D::Map (core/texture.cpp:93-99), MIPMap lookups (core/mipmap.h:203-338) -------------------------------
struct DevUvDiff { float dudx, dvdx, dudy, dvdy; };
__device__ __forceinline__ bool solve2x2(float a00, float a01, float a10, float a11, float b0, float b1, float *x0, float *x1) {   // core/transform.cpp:41-49
    const float det = a00 * a11 - a01 * a10;
    if (fabsf(det) < 1e-10f) return false;
    *x0 = (a11 * b0 - a01 * b1) / det;
    *x1 = (a00 * b1 - a10 * b0) / det;
    if (is_nan(*x0) || is_nan(*x1)) return false;
    return true;
}
__device__ __forceinline__ void compute_differentials(const DevSI &si, const DevTexGeom &tg, const DevRayDiff &rd, DevUvDiff *o) {
    o->dudx = o->dvdx = o->dudy = o->dvdy = 0.f;
    const vec3 n = si.n, p = si.p;
    const float d = dot(n, vec3(p.x, p.y, p.z));
    const float tx = -(dot(n, rd.rxO) - d) / dot(n, rd.rxD);
    if (is_inf(tx) || is_nan(tx)) return;
    const vec3 px = rd.rxO + tx * rd.rxD;
    const float ty = -(dot(n, rd.ryO) - d) / dot(n, rd.ryD);
    if (is_inf(ty) || is_nan(ty)) return;
    const vec3 py = rd.ryO + ty * rd.ryD;
    int d0, d1;
    if (fabsf(n.x) > fabsf(n.y) && fabsf(n.x) > fabsf(n.z)) { d0 = 1; d1 = 2; }
    else if (fabsf(n.y) > fabsf(n.z)) { d0 = 0; d1 = 2; }
    else { d0 = 0; d1 = 1; }
    const float a00 = tg.dpdu.get(d0), a01 = tg.dpdv.get(d0), a10 = tg.dpdu.get(d1), a11 = tg.dpdv.get(d1);
    if (!solve2x2(a00, a01, a10, a11, px.get(d0) - p.get(d0), px.get(d1) - p.get(d1), &o->dudx, &o->dvdx)) o->dudx = o->dvdx = 0.f;
    if (!solve2x2(a00, a01, a10, a11, py.get(d0) - p.get(d0), py.get(d1) - p.get(d1), &o->dudy, &o->dvdy)) o->dudy = o->dvdy = 0.f;
}
__device__ __forceinline__ int mod_i(int a, int b) { const int r = a - (a / b) * b; return r < 0 ? r + b : r; }
__device__ __forceinline__ rgb mip_texel(const DevScene &sc, const DevTexture &tx, int level, int s, int t) {
    const DevMipLevel l = sc.mipLevels[tx.firstLevel + level];
    if (tx.wrap == 0) { s = mod_i(s, l.w); t = mod_i(t, l.h); }
    else if (tx.wrap == 2) { s = s < 0 ? 0 : (s > l.w - 1 ? l.w - 1 : s); t = t < 0 ? 0 : (t > l.h - 1 ? l.h - 1 : t); }
    else if (s < 0 || s >= l.w || t < 0 || t >= l.h) return rgb(0.f);
    const float *p = sc.texels + l.offset + 3 * ((size_t)t * l.w + s);
    return rgb(p[0], p[1], p[2]);
}
__device__ __forceinline__ rgb mip_triangle(const DevScene &sc, const DevTexture &tx, int level, float su, float sv) {
    level = level < 0 ? 0 : (level > (int)tx.nLevels - 1 ? (int)tx.nLevels - 1 : level);
    const DevMipLevel l = sc.mipLevels[tx.firstLevel + level];
    const float s = su * l.w - 0.5f, t = sv * l.h - 0.5f;
    const int s0 = (int)floorf(s), t0 = (int)floorf(t);
    const float ds = s - s0, dt = t - t0;
    return (1 - ds) * (1 - dt) * mip_texel(sc, tx, level, s0, t0) + (1 - ds) * dt * mip_texel(sc, tx, level, s0, t0 + 1) +
           ds * (1 - dt) * mip_texel(sc, tx, level, s0 + 1, t0) + ds * dt * mip_texel(sc, tx, level, s0 + 1, t0 + 1);
}
__device__ __forceinline__ float log2_f(float x) { const float invLog2 = 1.442695040888963387004650940071; return det_logf(x) * invLog2; }   // core/pbrt.h:328-331
__device__ __forceinline__ rgb lerp_rgb(float t, rgb a, rgb b) { return (1 - t) * a + t * b; }
__device__ __forceinline__ rgb mip_ewa(const DevScene &sc, const DevTexture &tx, int level, float su, float sv, float d0x, float d0y, float d1x, float d1y) {
    if (level >= (int)tx.nLevels) return mip_texel(sc, tx, (int)tx.nLevels - 1, 0, 0);
    const DevMipLevel l = sc.mipLevels[tx.firstLevel + level];
    su = su * l.w - 0.5f; sv = sv * l.h - 0.5f;
    d0x *= l.w; d0y *= l.h; d1x *= l.w; d1y *= l.h;
    float A = d0y * d0y + d1y * d1y + 1;
    float B = -2 * (d0x * d0y + d1x * d1y);
    float C = d0x * d0x + d1x * d1x + 1;
    const float invF = 1 / (A * C - B * B * 0.25f);
    A *= invF; B *= invF; C *= invF;
    const float det = -B * B + 4 * A * C;
    const float invDet = 1 / det;
    const float uSqrt = sqrtf(det * C), vSqrt = sqrtf(A * det);
    const int s0 = (int)ceilf(su - 2 * invDet * uSqrt), s1 = (int)floorf(su + 2 * invDet * uSqrt);
    const int t0 = (int)ceilf(sv - 2 * invDet * vSqrt), t1 = (int)floorf(sv + 2 * invDet * vSqrt);
    rgb sum(0.f);
    float sumWts = 0;
    for (int it = t0; it <= t1; ++it) {
        const float tt = it - sv;
        for (int is = s0; is <= s1; ++is) {
            const float ss = is - su;
            const float r2 = A * ss * ss + B * ss * tt + C * tt * tt;
            if (r2 < 1) {
                int index = (int)(r2 * 128); if (index > 127) index = 127;
                const float weight = sc.weightLut[index];
                sum = sum + mip_texel(sc, tx, level, is, it) * weight;
                sumWts += weight;
            }
        }
    }
    return sum / sumWts;
}
// ImageTexture::Evaluate (textures/imagemap.h:82-89): UVMapping2D::Map, then MIPMap::Lookup(st, dstdx, dstdy)
__device__ __forceinline__ rgb eval_image_texture(const DevScene &sc, int texId, const DevTexGeom &tg, const DevUvDiff &uv) {
    const DevTexture tx = sc.textures[texId];
    float d0x = tx.su * uv.dudx, d0y = tx.sv * uv.dvdx, d1x = tx.su * uv.dudy, d1y = tx.sv * uv.dvdy;
    const float su = tx.su * tg.u + tx.du, sv = tx.sv * tg.v + tx.dv;
    if (tx.trilinear) {
        const float width = 2 * sel_max(sel_max(fabsf(d0x), fabsf(d0y)), sel_max(fabsf(d1x), fabsf(d1y)));
        const float level = (int)tx.nLevels - 1 + log2_f(sel_max(width, (float)1e-8));
        if (level < 0) return mip_triangle(sc, tx, 0, su, sv);
        else if (level >= (int)tx.nLevels - 1) return mip_texel(sc, tx, (int)tx.nLevels - 1, 0, 0);
        const int iLevel = (int)floorf(level);
        const float delta = level - iLevel;
        return lerp_rgb(delta, mip_triangle(sc, tx, iLevel, su, sv), mip_triangle(sc, tx, iLevel + 1, su, sv));
    }
    if (d0x * d0x + d0y * d0y < d1x * d1x + d1y * d1y) { float t; t = d0x; d0x = d1x; d1x = t; t = d0y; d0y = d1y; d1y = t; }
    const float majorLength = sqrtf(d0x * d0x + d0y * d0y);
    float minorLength = sqrtf(d1x * d1x + d1y * d1y);
    if (minorLength * tx.maxAniso < majorLength && minorLength > 0) {
        const float scale = majorLength / (minorLength * tx.maxAniso);
        d1x *= scale; d1y *= scale;
        minorLength *= scale;
    }
    if (minorLength == 0) return mip_triangle(sc, tx, 0, su, sv);
    const float lod = sel_max((float)0, (int)tx.nLevels - (float)1 + log2_f(minorLength));
    const int ilod = (int)floorf(lod);
    rgb a(0.f), b(0.f);      // the two levels through one copy of the filter loop (the body is inlined: a call here spills the caller's live state to scratch)
#pragma unroll 1
    for (int k = 0; k < 2; ++k) {
        const rgb r = mip_ewa(sc, tx, ilod + k, su, sv, d0x, d0y, d1x, d1y);
        if (k == 0) a = r; else b = r;
    }
    return lerp_rgb(lod - ilod, a, b);
}

// kdOverride / ksOverride: evaluated image textures of the material's Kd / Ks (null: the constants)
// opOverride: the evaluated image texture of an uber material's opacity
__device__ __forceinline__ void bsdf_init(const DevScene &sc, const DevSI &si, DevBsdf *b, const rgb *kdOverride = nullptr,
                                          const rgb *ksOverride = nullptr, const rgb *opOverride = nullptr) {
    b->ns = si.ns; b->ng = si.n;
    b->ss = normalize(si.sdpdu);
    b->ts = cross(b->ns, b->ss);
    b->alpha = 0; b->hasD = false; b->hasS = false; b->Rd = rgb(0.f); b->Rs = rgb(0.f);
    b->hasR = false; b->hasT = false; b->Rr = rgb(0.f); b->oren = false; b->orenA = 1.f; b->orenB = 0.f;
    const DevMaterial m = sc.materials[sc.shapes[si.shape].material];
    b->alphaY = 0; b->kind = 0; b->eta = 1.f; b->frI = 1.5f; b->frT = 1.f; b->uber = nullptr;
    if (m.type == 6) {      // UberMaterial, materials/uber.cpp:45-108: Lambertian + microfacet (FresnelDielectric(1, e)) as the plastic pair, and up to
                            // three specular lobes (1 - opacity straight through, Kr, Kt) that only the next-segment sampling sees (bsdf_sample all = true)
        const rgb op = clamp0(opOverride ? *opOverride : rgb(m.opacity[0], m.opacity[1], m.opacity[2]));
        const rgb t = clamp0(-op + rgb(1.f));
        b->kind = 5; b->uber = &sc.materials[sc.shapes[si.shape].material]; b->op = op;
        b->eta = is_black(t) ? m.eta : 1.f;
        const rgb kd = op * clamp0(kdOverride ? *kdOverride : rgb(m.Kd[0], m.Kd[1], m.Kd[2]));
        if (!is_black(kd)) { b->hasD = true; b->Rd = kd; }
        const rgb ks = op * clamp0(ksOverride ? *ksOverride : rgb(m.Ks[0], m.Ks[1], m.Ks[2]));
        if (!is_black(ks)) { b->hasS = true; b->Rs = ks; b->alpha = m.alpha; b->alphaY = m.alphaY; b->frI = 1.f; b->frT = m.eta; }
        return;
    }
    if (m.type == 5) {      // GlassMaterial, smooth (materials/glass.cpp:44-65 with allowMultipleLobes): one FresnelSpecular lobe
        b->eta = m.alpha;
        const rgb R = clamp0(ksOverride ? *ksOverride : rgb(m.Ks[0], m.Ks[1], m.Ks[2])), T = clamp0(kdOverride ? *kdOverride : rgb(m.Kd[0], m.Kd[1], m.Kd[2]));
        if (is_black(R) && is_black(T)) return;
        if (m.roughGlass) {      // rough dielectric (:66-93): MicrofacetReflection(R, FresnelDielectric(1, eta)) then MicrofacetTransmission(T, 1, eta)
            b->kind = 6; b->alpha = m.Kr[0]; b->alphaY = m.Kr[1]; b->frI = 1.f; b->frT = b->eta;
            if (!is_black(R)) { b->hasS = true; b->Rs = R; }
            if (!is_black(T)) { b->hasT = true; b->Rd = T; }
            return;
        }
        b->hasR = true; b->kind = 4; b->Rr = R; b->Rd = T;
        return;
    }
    if (m.type == 3) {      // SubstrateMaterial, materials/substrate.cpp:44-65: one FresnelBlend lobe unless both reflectances are black
        const rgb d = clamp0(kdOverride ? *kdOverride : rgb(m.Kd[0], m.Kd[1], m.Kd[2])), sp = clamp0(ksOverride ? *ksOverride : rgb(m.Ks[0], m.Ks[1], m.Ks[2]));
        if (!is_black(d) || !is_black(sp)) { b->kind = 2; b->hasS = true; b->Rd = d; b->Rs = sp; b->alpha = m.alpha; b->alphaY = m.alphaY; }
        return;
    }
    if (m.type == 4) {      // MetalMaterial, materials/metal.cpp:59-79: one conductor microfacet lobe, always
        b->kind = 3; b->hasS = true; b->Rd = rgb(m.Kd[0], m.Kd[1], m.Kd[2]); b->Rs = rgb(m.Ks[0], m.Ks[1], m.Ks[2]); b->alpha = m.alpha; b->alphaY = m.alphaY;
        return;
    }
    if (m.type == 2) {      // MirrorMaterial, materials/mirror.cpp:44-56
        rgb kr = clamp0(ksOverride ? *ksOverride : rgb(m.Ks[0], m.Ks[1], m.Ks[2]));
        if (!is_black(kr)) { b->hasR = true; b->Rr = kr; }
        return;
    }
    rgb kd = clamp0(kdOverride ? *kdOverride : rgb(m.Kd[0], m.Kd[1], m.Kd[2]));
    if (!is_black(kd)) { b->hasD = true; b->Rd = kd; if (m.oren) { b->oren = true; b->orenA = m.orenA; b->orenB = m.orenB; } }
    if (m.type == 1) {
        rgb ks = clamp0(ksOverride ? *ksOverride : rgb(m.Ks[0], m.Ks[1], m.Ks[2]));
        if (!is_black(ks)) { b->hasS = true; b->Rs = ks; b->alpha = m.alpha; b->alphaY = m.alpha; }
    }
}
__device__ __forceinline__ vec3 to_local(const DevBsdf &b, vec3 v) { return vec3(dot(v, b.ss), dot(v, b.ts), dot(v, b.ns)); }
__device__ __forceinline__ vec3 to_world(const DevBsdf &b, vec3 v) {
    return vec3(b.ss.x * v.x + b.ts.x * v.y + b.ns.x * v.z, b.ss.y * v.x + b.ts.y * v.y + b.ns.y * v.z,
                b.ss.z * v.x + b.ts.z * v.y + b.ns.z * v.z);
}
// Both lobe types (REFLECTION|DIFFUSE, REFLECTION|GLOSSY) match BSDF_ALL and
// BSDF_ALL & ~BSDF_SPECULAR, the only flag sets PathIntegrator passes (path.cpp:129,144;
// integrator.cpp:114), so NumComponents(flags) is the lobe count.
__device__ __forceinline__ int bsdf_num(const DevBsdf &b) { return (b.hasD ? 1 : 0) + (b.hasS ? 1 : 0) + (b.hasT ? 1 : 0); }
// BSDF::f, core/reflection.cpp:670-684 (all lobes here are reflective)
__device__ __forceinline__ rgb bsdf_f(const DevBsdf &b, vec3 woW, vec3 wiW) {
    vec3 wi = to_local(b, wiW), wo = to_local(b, woW);
    if (wo.z == 0) return rgb(0.f);
    bool reflect = dot(wiW, b.ng) * dot(woW, b.ng) > 0;
    rgb f(0.f);
    if (b.hasD && reflect) f = f + lambert_f(b, wo, wi);
    if (b.hasS && reflect) f = f + (b.kind == 2 ? blend_f(b, wo, wi) : mf_f(b, wo, wi));
    if (b.hasT && !reflect) f = f + mt_f(b, wo, wi);      // (the one transmissive non-specular lobe: rough glass)
    return f;
}
// BSDF::Pdf, core/reflection.cpp:764-778
__device__ __forceinline__ float bsdf_pdf(const DevBsdf &b, vec3 woW, vec3 wiW) {
    const int matching = bsdf_num(b);
    if (matching == 0) return 0.f;
    vec3 wo = to_local(b, woW), wi = to_local(b, wiW);
    if (wo.z == 0) return 0.f;
    float pdf = 0.f;
    if (b.hasD) pdf += lambert_pdf(wo, wi);
    if (b.hasS) pdf += b.kind == 2 ? blend_pdf(b, wo, wi) : mf_pdf(b, wo, wi);
    if (b.hasT) pdf += mt_pdf(b, wo, wi);
    return pdf / matching;
}
// BSDF::Sample_f, core/reflection.cpp:703-762.  *pdf keeps its incoming value on the
// "wo.z == 0" early return, as in the reference.
// all: Sample_f(BSDF_ALL) of the path's next segment (integrators/path.cpp:147-148); false: EstimateDirect's Sample_f(BSDF_ALL & ~BSDF_SPECULAR).
// They differ only for UberMaterial, the one BSDF here that mixes specular and non-specular lobes.
__device__ __forceinline__ rgb bsdf_sample(const DevBsdf &b, vec3 woW, vec3 *wiW, float u0, float u1, float *pdf, int *sampledType, bool all = false) {
    int nBefore = 0, nAfter = 0;      // uber's specular lobes in front of / behind the Lambertian + microfacet pair, in the order uber.cpp adds them
    rgb uT0(0.f), uKr(0.f), uKt(0.f);
    if (all && b.kind == 5) {
        const DevMaterial &m = *b.uber;
        const rgb op = b.op;
        uT0 = clamp0(-op + rgb(1.f));
        uKr = op * clamp0(rgb(m.Kr[0], m.Kr[1], m.Kr[2])); uKt = op * clamp0(rgb(m.Kt[0], m.Kt[1], m.Kt[2]));
        nBefore = is_black(uT0) ? 0 : 1;
        nAfter = (is_black(uKr) ? 0 : 1) + (is_black(uKt) ? 0 : 1);
    }
    if (nBefore + nAfter > 0) {
        const int matchingAll = nBefore + bsdf_num(b) + nAfter;
        const int compAll = sel_min((int)floorf(u0 * matchingAll), matchingAll - 1);
        const bool pickT0 = nBefore == 1 && compAll == 0;
        const int afterIdx = compAll - nBefore - bsdf_num(b);      // >= 0: a lobe behind the pair
        if (pickT0 || afterIdx >= 0) {
            const vec3 wo = to_local(b, woW);
            if (wo.z == 0) return rgb(0.f);
            *pdf = 0;
            const bool pickR = !pickT0 && !is_black(uKr) && afterIdx == 0;
            const float e = b.uber->eta;
            if (pickR) {      // SpecularReflection(kr, FresnelDielectric(1, e)), core/reflection.cpp:136-143
                const vec3 wi(-wo.x, -wo.y, wo.z);
                *sampledType = BX_REFLECTION | BX_SPECULAR;
                *pdf = 1.f / matchingAll;      // (pdf = 1, then "/= matchingComps", :748)
                *wiW = to_world(b, wi);
                return rgb(fr_dielectric(cos_theta(wi), 1.f, e)) * uKr / abs_cos_theta(wi);
            }
            // SpecularTransmission(T, etaA = 1, etaB), core/reflection.cpp:145-163 (TransportMode::Radiance): T0 = 1 - opacity with etaB = 1, or kt with etaB = e
            const rgb T = pickT0 ? uT0 : uKt;
            const float etaB = pickT0 ? 1.f : e;
            const bool entering = cos_theta(wo) > 0;
            const float etaI = entering ? 1.f : etaB, etaT = entering ? etaB : 1.f;
            const vec3 n = face_forward(vec3(0.f, 0.f, 1.f), wo);
            const float er = etaI / etaT;
            const float cosThetaI = dot(n, wo);
            const float sin2ThetaI = sel_max(0.f, 1 - cosThetaI * cosThetaI);
            const float sin2ThetaT = er * er * sin2ThetaI;
            if (sin2ThetaT >= 1) { *sampledType = 0; return rgb(0.f); }      // Refract() failed: BSDF::Sample_f sees pdf == 0
            const float cosThetaT = sqrtf(1 - sin2ThetaT);
            const vec3 wi = er * -wo + (er * cosThetaI - cosThetaT) * n;
            rgb ft = T * (rgb(1.f) - rgb(fr_dielectric(cos_theta(wi), 1.f, etaB)));
            ft = ft * ((etaI * etaI) / (etaT * etaT));
            *sampledType = BX_SPECULAR | BX_TRANSMISSION;
            *pdf = 1.f / matchingAll;
            *wiW = to_world(b, wi);
            return ft / abs_cos_theta(wi);
        }
    }
    if (b.hasR) {
        // the one lobe is SpecularReflection (core/reflection.cpp:136-143; FresnelNoOp::Evaluate == Spectrum(1.)): BSDF::Sample_f
        // returns its value as sampled, with its pdf of 1 (:744-760 skip specular lobes)
        const vec3 wo = to_local(b, woW);
        if (wo.z == 0) return rgb(0.f);
        if (b.kind == 4) {      // FresnelSpecular::Sample_f, core/reflection.cpp:477-512 (etaA = 1, etaB = eta, TransportMode::Radiance)
            const float uu = sel_min(u0, HPRT_ONE_MINUS_EPS);      // uRemapped of the one matching lobe
            const float F = fr_dielectric(cos_theta(wo), 1.f, b.eta);
            if (uu < F) {
                const vec3 wi(-wo.x, -wo.y, wo.z);
                *sampledType = BX_SPECULAR | BX_REFLECTION;
                *pdf = F;
                *wiW = to_world(b, wi);
                return b.Rr * F / abs_cos_theta(wi);
            }
            const bool entering = cos_theta(wo) > 0;
            const float etaI = entering ? 1.f : b.eta, etaT = entering ? b.eta : 1.f;
            // Refract(wo, Faceforward(Normal3f(0, 0, 1), wo), etaI / etaT, wi), core/reflection.h:96-108
            const vec3 n = face_forward(vec3(0.f, 0.f, 1.f), wo);
            const float er = etaI / etaT;
            const float cosThetaI = dot(n, wo);
            const float sin2ThetaI = sel_max(0.f, 1 - cosThetaI * cosThetaI);
            const float sin2ThetaT = er * er * sin2ThetaI;
            *pdf = 0;
            if (sin2ThetaT >= 1) { *sampledType = 0; return rgb(0.f); }      // total internal reflection: BSDF::Sample_f sees pdf == 0
            const float cosThetaT = sqrtf(1 - sin2ThetaT);
            const vec3 wi = er * -wo + (er * cosThetaI - cosThetaT) * n;
            rgb ft = b.Rd * (1 - F);
            ft = ft * ((etaI * etaI) / (etaT * etaT));
            *sampledType = BX_SPECULAR | BX_TRANSMISSION;
            *pdf = 1 - F;
            *wiW = to_world(b, wi);
            return ft / abs_cos_theta(wi);
        }
        const vec3 wi(-wo.x, -wo.y, wo.z);
        *pdf = 1;
        *sampledType = BX_REFLECTION | BX_SPECULAR;
        *wiW = to_world(b, wi);
        return rgb(1.f) * b.Rr / abs_cos_theta(wi);
    }
    if (b.kind == 6) {      // rough glass: BSDF::Sample_f over {MicrofacetReflection, MicrofacetTransmission} (core/reflection.cpp:703-762, 402-414, 425-435)
        const int matching6 = (b.hasS ? 1 : 0) + (b.hasT ? 1 : 0);
        if (matching6 == 0) { *pdf = 0; *sampledType = 0; return rgb(0.f); }
        const int comp6 = sel_min((int)floorf(u0 * matching6), matching6 - 1);
        const bool pickT = b.hasS ? comp6 == 1 : true;
        const float ur0 = sel_min(u0 * matching6 - comp6, HPRT_ONE_MINUS_EPS);
        const vec3 wo = to_local(b, woW);
        if (wo.z == 0) return rgb(0.f);
        *pdf = 0;
        *sampledType = pickT ? (BX_TRANSMISSION | BX_GLOSSY) : (BX_REFLECTION | BX_GLOSSY);
        const vec3 wh = tr_sample_wh(b.alpha, b.alphaY, wo, ur0, u1);
        vec3 wi;
        if (!pickT) {
            wi = -wo + 2 * dot(wo, wh) * wh;
            if (same_hemisphere(wo, wi)) *pdf = tr_pdf(b.alpha, b.alphaY, wo, wh) / (4 * dot(wo, wh));
        } else {
            // Refract(wo, (Normal3f)wh, eta, &wi), core/reflection.h:96-108
            const float er = cos_theta(wo) > 0 ? (1.f / b.eta) : (b.eta / 1.f);
            const float cosThetaI = dot(wh, wo);
            const float sin2ThetaI = sel_max(0.f, 1 - cosThetaI * cosThetaI);
            const float sin2ThetaT = er * er * sin2ThetaI;
            if (!(sin2ThetaT >= 1)) {
                const float cosThetaT = sqrtf(1 - sin2ThetaT);
                wi = er * -wo + (er * cosThetaI - cosThetaT) * wh;
                *pdf = mt_pdf(b, wo, wi);
            }
        }
        if (*pdf == 0) { *sampledType = 0; return rgb(0.f); }
        *wiW = to_world(b, wi);
        if (matching6 > 1) { *pdf += pickT ? mf_pdf(b, wo, wi) : mt_pdf(b, wo, wi); *pdf /= matching6; }
        const bool reflect6 = dot(*wiW, b.ng) * dot(woW, b.ng) > 0;
        rgb f6(0.f);
        if (b.hasS && reflect6) f6 = f6 + mf_f(b, wo, wi);
        if (b.hasT && !reflect6) f6 = f6 + mt_f(b, wo, wi);
        return f6;
    }
    const int nPair = bsdf_num(b);
    const int matching = nBefore + nPair + nAfter;      // (nBefore = nAfter = 0 unless an uber BSDF is sampled over all its lobes)
    if (matching == 0) { *pdf = 0; *sampledType = 0; return rgb(0.f); }
    const int compAll = sel_min((int)floorf(u0 * matching), matching - 1);
    const int comp = compAll - nBefore;                 // within the Lambertian / microfacet pair (the specular picks returned above)
    const bool pickSpec = (nPair == 2) ? (comp == 1) : b.hasS;
    const float ur0 = sel_min(u0 * matching - compAll, HPRT_ONE_MINUS_EPS);
    vec3 wi, wo = to_local(b, woW);
    if (wo.z == 0) return rgb(0.f);
    *pdf = 0;
    *sampledType = pickSpec ? (BX_REFLECTION | BX_GLOSSY) : (BX_REFLECTION | BX_DIFFUSE);
    if (!pickSpec) {      // BxDF::Sample_f: cosine-weighted hemisphere
        float dx, dy; concentric_disk(ur0, u1, &dx, &dy);
        float z = sqrtf(sel_max(0.f, 1 - dx * dx - dy * dy));
        wi = vec3(dx, dy, z);
        if (wo.z < 0) wi.z *= -1;
        *pdf = lambert_pdf(wo, wi);
    } else if (b.kind == 2) {      // FresnelBlend::Sample_f, core/reflection.cpp:450-468
        float v0 = ur0;
        if ((double)v0 < .5) {
            v0 = sel_min(2 * v0, HPRT_ONE_MINUS_EPS);
            float dx, dy; concentric_disk(v0, u1, &dx, &dy);
            float z = sqrtf(sel_max(0.f, 1 - dx * dx - dy * dy));
            wi = vec3(dx, dy, z);
            if (wo.z < 0) wi.z *= -1;
            *pdf = blend_pdf(b, wo, wi);
        } else {
            v0 = sel_min(2 * (v0 - .5f), HPRT_ONE_MINUS_EPS);
            vec3 wh = tr_sample_wh(b.alpha, b.alphaY, wo, v0, u1);
            wi = -wo + 2 * dot(wo, wh) * wh;
            if (same_hemisphere(wo, wi)) *pdf = blend_pdf(b, wo, wi);
        }
    } else {              // MicrofacetReflection::Sample_f, core/reflection.cpp:402-414 (wo.z != 0 here)
        vec3 wh = tr_sample_wh(b.alpha, b.alphaY, wo, ur0, u1);
        wi = -wo + 2 * dot(wo, wh) * wh;
        if (same_hemisphere(wo, wi)) *pdf = tr_pdf(b.alpha, b.alphaY, wo, wh) / (4 * dot(wo, wh));
    }
    if (*pdf == 0) { *sampledType = 0; return rgb(0.f); }
    *wiW = to_world(b, wi);
    if (nPair > 1) *pdf += pickSpec ? lambert_pdf(wo, wi) : mf_pdf(b, wo, wi);      // (the other matching lobes; specular ones have Pdf() == 0)
    if (matching > 1) *pdf /= matching;
    bool reflect = dot(*wiW, b.ng) * dot(woW, b.ng) > 0;
    rgb f(0.f);
    if (b.hasD && reflect) f = f + lambert_f(b, wo, wi);
    if (b.hasS && reflect) f = f + (b.kind == 2 ? blend_f(b, wo, wi) : mf_f(b, wo, wi));
    return f;
}

// ---------------------------------------------------------------------------
// Lights
// ---------------------------------------------------------------------------
struct DevIt { vec3 p, pErr, n; };   // Interaction subset

__device__ __forceinline__ vec3 spherical_dir(float sinTheta, float cosTheta, float phi, vec3 x, vec3 y, vec3 z) {
    float sn, cs; det_sincosf(phi, &sn, &cs);
    return sinTheta * cs * x + sinTheta * sn * y + cosTheta * z;
}
// Sphere::Sample(u) (shapes/sphere.cpp:217-230)
__device__ __forceinline__ DevIt sphere_sample_area(const DevSphere &s, bool reverse, float u0, float u1, float *pdf) {
    float z = 1 - 2 * u0;
    float r = sqrtf(sel_max(0.f, 1.0f - z * z));
    float phi = 2 * HPRT_PI * u1;
    float sn, cs; det_sincosf(phi, &sn, &cs);
    vec3 pObj = vec3(0, 0, 0) + s.radius * vec3(r * cs, r * sn, z);
    DevIt it;
    it.n = normalize(xf_normal(s.w2o, pObj));
    if (reverse) it.n = it.n * -1.f;
    pObj = pObj * (s.radius / dist(pObj, vec3(0, 0, 0)));
    vec3 pObjErr = gamma_n(5) * vabs(pObj);
    it.p = xf_point_err_in(s.o2w, pObj, pObjErr, &it.pErr);
    *pdf = 1 / (s.phiMax * s.radius * (s.zMax - s.zMin));
    return it;
}
// The "reference point inside the sphere" test shared by Sphere::Sample(ref,u) and
// Sphere::Pdf (shapes/sphere.cpp:236-239, 297-300): same operands, same result.
__device__ __forceinline__ bool sphere_ref_inside(const DevSphere &s, const DevIt &ref) {
    vec3 pCenter = xf_point(s.o2w, vec3(0, 0, 0));
    vec3 pOrigin = offset_ray_origin(ref.p, ref.pErr, ref.n, pCenter - ref.p);
    return dist2(pOrigin, pCenter) <= s.radius * s.radius;
}
// Sphere::Sample(ref, u) (shapes/sphere.cpp:232-292).  INSIDE_POSSIBLE = false is used by
// callers that have already established sphere_ref_inside() == false.
template <bool INSIDE_POSSIBLE>
__device__ __forceinline__ DevIt sphere_sample(const DevSphere &s, bool reverse, const DevIt &ref, float u0, float u1, float *pdf) {
    vec3 pCenter = xf_point(s.o2w, vec3(0, 0, 0));
    if (INSIDE_POSSIBLE && sphere_ref_inside(s, ref)) {
        DevIt intr = sphere_sample_area(s, reverse, u0, u1, pdf);
        vec3 wi = intr.p - ref.p;
        if (length2(wi) == 0) *pdf = 0;
        else { wi = normalize(wi); *pdf *= dist2(ref.p, intr.p) / absdot(intr.n, -wi); }
        if (is_inf(*pdf)) *pdf = 0.f;
        return intr;
    }
    vec3 wc = normalize(pCenter - ref.p), wcX, wcY;
    coordinate_system(wc, &wcX, &wcY);
    float sinThetaMax2 = s.radius * s.radius / dist2(ref.p, pCenter);
    float cosThetaMax = sqrtf(sel_max(0.f, 1 - sinThetaMax2));
    float cosTheta = (1 - u0) + u0 * cosThetaMax;
    float sinTheta = sqrtf(sel_max(0.f, 1 - cosTheta * cosTheta));
    float phi = u1 * 2 * HPRT_PI;
    float dc = dist(ref.p, pCenter);
    float ds = dc * cosTheta - sqrtf(sel_max(0.f, s.radius * s.radius - dc * dc * sinTheta * sinTheta));
    float cosAlpha = (dc * dc + s.radius * s.radius - ds * ds) / (2 * dc * s.radius);
    float sinAlpha = sqrtf(sel_max(0.f, 1 - cosAlpha * cosAlpha));
    vec3 nWorld = spherical_dir(sinAlpha, cosAlpha, phi, -wcX, -wcY, -wc);
    vec3 pWorld = pCenter + s.radius * vec3(nWorld.x, nWorld.y, nWorld.z);
    DevIt it;
    it.p = pWorld;
    it.pErr = gamma_n(5) * vabs(pWorld);
    it.n = nWorld;
    if (reverse) it.n = it.n * -1.f;
    *pdf = 1 / (2 * HPRT_PI * (1 - cosThetaMax));
    return it;
}
__device__ __forceinline__ rgb area_L(const DevLight &l, vec3 n, vec3 w) {   // lights/diffuse.h:56-58
    return (l.twoSided || dot(n, w) > 0) ? rgb(l.I[0], l.I[1], l.I[2]) : rgb(0.f);
}
// GeometricPrimitive::GetAreaLight of an ordered primitive: the sphere's light, or the light of this one triangle (aux - 1)
__device__ __forceinline__ int prim_area_light(const DevScene &sc, int32_t prim) {
    const uint32_t tag = __float_as_uint(sc.tris[3 * prim].w);
    if ((tag & TAG_KIND_MASK) == 0u) return (int)__float_as_uint(sc.tris[3 * prim + 2].w) - 1;
    return sc.shapes[__float_as_uint(sc.tris[3 * prim + 1].w)].areaLight;
}
// Triangle::Area, shapes/triangle.cpp:576-582 (0.5 is a double literal)
__device__ __forceinline__ float triangle_area(vec3 p0, vec3 p1, vec3 p2) {
    return (float)(0.5 * (double)length(cross(p1 - p0, p2 - p0)));
}
// Shape::Sample(ref, u, pdf) (core/shape.cpp:55-70) over Triangle::Sample(u, pdf) (shapes/triangle.cpp:596-621)
__device__ __forceinline__ DevIt triangle_sample(const DevScene &sc, int32_t prim, const DevIt &ref, float u0, float u1, float *pdf) {
    const float4 v0 = sc.tris[3 * prim], v1 = sc.tris[3 * prim + 1], v2 = sc.tris[3 * prim + 2];
    const vec3 p0(v0.x, v0.y, v0.z), p1(v1.x, v1.y, v1.z), p2(v2.x, v2.y, v2.z);
    const DevShape sh = sc.shapes[__float_as_uint(v1.w)];
    const float su0 = sqrtf(u0);                       // UniformSampleTriangle, core/sampling.cpp:154-157
    const float b0 = 1 - su0, b1 = u1 * su0;
    DevIt it;
    it.p = b0 * p0 + b1 * p1 + (1 - b0 - b1) * p2;
    it.n = normalize(cross(p1 - p0, p2 - p0));
    if (sh.flags & SHAPE_HAS_N) {
        const float4 m0 = sc.primN[3 * prim], m1 = sc.primN[3 * prim + 1], m2 = sc.primN[3 * prim + 2];
        const vec3 ns = b0 * vec3(m0.x, m0.y, m0.z) + b1 * vec3(m1.x, m1.y, m1.z) + (1 - b0 - b1) * vec3(m2.x, m2.y, m2.z);
        it.n = face_forward(it.n, ns);
    } else if (sh.flags & SHAPE_FLIP) it.n = it.n * -1.f;
    const vec3 pAbsSum = vabs(b0 * p0) + vabs(b1 * p1) + vabs((1 - b0 - b1) * p2);
    it.pErr = gamma_n(6) * pAbsSum;
    *pdf = 1 / triangle_area(p0, p1, p2);
    vec3 wi = it.p - ref.p;
    if (length2(wi) == 0) *pdf = 0;
    else {
        wi = normalize(wi);
        *pdf *= dist2(ref.p, it.p) / absdot(it.n, -wi);
        if (is_inf(*pdf)) *pdf = 0.f;
    }
    return it;
}
// Would Triangle::Intersect of light triangle `prim` accept this ray for SOME tMax?  (The test the traversal runs, with
// tMax = infinity: a smaller tMax only rejects more.)  false => the triangle can never be the ray's closest hit.
__device__ __forceinline__ bool triangle_may_hit(const DevScene &sc, int32_t prim, const DRay &r, float *b0, float *b1, float *b2, float *t) {
    const float4 v0 = sc.tris[3 * prim], v1 = sc.tris[3 * prim + 1], v2 = sc.tris[3 * prim + 2];
    if (__float_as_uint(v0.w) & TAG_BOGUS) return false;      // Triangle::Intersect returns false on it (shapes/triangle.cpp:309-316)
    const RayShear sh = ray_shear(r.d);
    return tri_test(vec3(v0.x, v0.y, v0.z), vec3(v1.x, v1.y, v1.z), vec3(v2.x, v2.y, v2.z), r.o, r.tMax, sh, b0, b1, b2, t);
}
// ---- InfiniteAreaLight (lights/infinite.cpp) -----------------------------------------------------------------------------
// Distribution1D::SampleContinuous (core/sampling.h:85-103) over a cdf / func pair of n entries
__device__ __forceinline__ float dist1d_sample_continuous(const float *cdf, const float *func, float funcInt, int n, float u, float *pdf, int *off) {
    const int size = n + 1;
    int first = 0, len = size;      // FindInterval, core/pbrt.h:403-415
    while (len > 0) {
        const int half = len >> 1, middle = first + half;
        if (cdf[middle] <= u) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    int offset = first - 1;
    offset = offset < 0 ? 0 : (offset > size - 2 ? size - 2 : offset);
    *off = offset;
    float du = u - cdf[offset];
    if ((cdf[offset + 1] - cdf[offset]) > 0) du /= (cdf[offset + 1] - cdf[offset]);
    *pdf = (funcInt > 0) ? func[offset] / funcInt : 0;
    return (offset + du) / n;
}
struct DevEnvTables { const float *condFunc, *condCdf, *condInt, *margCdf; };
__device__ __forceinline__ DevEnvTables env_tables(const DevScene &sc, const DevEnvLight &e) {
    DevEnvTables t;
    t.condFunc = sc.envData + e.off; t.condCdf = t.condFunc + (size_t)e.nv * e.nu; t.condInt = t.condCdf + (size_t)e.nv * (e.nu + 1); t.margCdf = t.condInt + e.nv;
    return t;
}
// Distribution2D::SampleContinuous / Pdf (core/sampling.h:132-145)
__device__ __forceinline__ void env_sample_uv(const DevScene &sc, const DevEnvLight &e, float u0, float u1, float *su, float *sv, float *pdf) {
    const DevEnvTables t = env_tables(sc, e);
    float pdf0, pdf1; int v, dummy;
    const float d1 = dist1d_sample_continuous(t.margCdf, t.condInt, e.margFuncInt, e.nv, u1, &pdf1, &v);
    const float d0 = dist1d_sample_continuous(t.condCdf + (size_t)v * (e.nu + 1), t.condFunc + (size_t)v * e.nu, t.condInt[v], e.nu, u0, &pdf0, &dummy);
    *pdf = pdf0 * pdf1;
    *su = d0; *sv = d1;
}
__device__ __forceinline__ float env_pdf_uv(const DevScene &sc, const DevEnvLight &e, float su, float sv) {
    const DevEnvTables t = env_tables(sc, e);
    int iu = (int)(su * e.nu), iv = (int)(sv * e.nv);
    iu = iu < 0 ? 0 : (iu > e.nu - 1 ? e.nu - 1 : iu); iv = iv < 0 ? 0 : (iv > e.nv - 1 ? e.nv - 1 : iv);
    return t.condFunc[(size_t)iv * e.nu + iu] / e.margFuncInt;
}
// Lmap->Lookup(st): MIPMap::Lookup(st, width = 0) — level = Levels() - 1 + Log2(1e-8) < 0 for every pyramid of at most 26 levels
// (checked at scene creation), i.e. the bilinear "triangle" filter at level 0 (core/mipmap.h:203-221)
__device__ __forceinline__ rgb env_lookup(const DevScene &sc, const DevEnvLight &e, float su, float sv) { return mip_triangle(sc, sc.textures[e.tex], 0, su, sv); }
// InfiniteAreaLight::Le(ray), lights/infinite.cpp:93-97 (SphericalPhi / SphericalTheta, core/geometry.h:1816-1824)
__device__ __forceinline__ rgb env_Le(const DevScene &sc, const DevEnvLight &e, vec3 d) {
    const vec3 w = normalize(xf_vector(e.w2l, d));
    float p = det_atan2f(w.y, w.x);
    const float phi = (p < 0) ? (p + 2 * HPRT_PI) : p;
    const float theta = det_acosf(clampf(w.z, -1.f, 1.f));
    return env_lookup(sc, e, phi * HPRT_INV_2PI, theta * HPRT_INV_PI);
}
// Light::Sample_Li (lights/point.cpp:44-53, distant.cpp:49-59, diffuse.cpp:68-81).  GENERIC: the generic shading variant, the
// only one that carries the code for shading points inside a sphere emitter and for triangle emitters.
template <bool GENERIC>
__device__ __forceinline__ rgb light_sample(const DevScene &sc, const DevLight &l, const DevIt &ref, float u0, float u1, vec3 *wi,
                                            float *pdf, DevIt *pLight) {
    vec3 lp(l.pos[0], l.pos[1], l.pos[2]);
    if (l.type == 0) {
        *wi = normalize(lp - ref.p);
        *pdf = 1.f;
        pLight->p = lp; pLight->pErr = vec3(); pLight->n = vec3();
        return rgb(l.I[0], l.I[1], l.I[2]) / dist2(lp, ref.p);
    } else if (l.type == 1) {
        *wi = lp;
        *pdf = 1;
        pLight->p = ref.p + lp * (2 * sc.worldRadius); pLight->pErr = vec3(); pLight->n = vec3();
        return rgb(l.I[0], l.I[1], l.I[2]);
    }
    if (GENERIC && l.type == 4) {      // lights/infinite.cpp:99-124
        const DevEnvLight &e = sc.envLights[l.shape];
        float su, sv, mapPdf;
        env_sample_uv(sc, e, u0, u1, &su, &sv, &mapPdf);
        if (mapPdf == 0) { *pdf = 0; return rgb(0.f); }
        const float theta = sv * HPRT_PI, phi = su * 2 * HPRT_PI;
        float sinTheta, cosTheta, sinPhi, cosPhi;
        det_sincosf(theta, &sinTheta, &cosTheta); det_sincosf(phi, &sinPhi, &cosPhi);
        *wi = xf_vector(e.l2w, vec3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta));
        *pdf = mapPdf / (2 * HPRT_PI * HPRT_PI * sinTheta);
        if (sinTheta == 0) *pdf = 0;
        pLight->p = ref.p + *wi * (2 * sc.worldRadius); pLight->pErr = vec3(); pLight->n = vec3();
        return env_lookup(sc, e, su, sv);
    }
    DevIt ps;
    if (GENERIC && l.type == 3) ps = triangle_sample(sc, l.prim, ref, u0, u1, pdf);
    else ps = sphere_sample<GENERIC>(sc.spheres[l.sphere], (l.shapeFlags & SHAPE_REVERSE) != 0, ref, u0, u1, pdf);
    if (*pdf == 0 || length2(ps.p - ref.p) == 0) { *pdf = 0; return rgb(0.f); }
    *wi = normalize(ps.p - ref.p);
    *pLight = ps;
    return area_L(l, ps.n, -*wi);
}
// Light::Pdf_Li for area lights = Sphere::Pdf (shapes/sphere.cpp:294-306; inside case and triangles:
// Shape::Pdf, core/shape.cpp:72-88)
template <bool GENERIC>
__device__ __forceinline__ float light_pdf(const DevScene &sc, const DevLight &l, const DevIt &ref, vec3 wi) {
    if (l.type < 2) return 0;
    if (GENERIC && l.type == 4) {      // lights/infinite.cpp:126-134
        const DevEnvLight &e = sc.envLights[l.shape];
        const vec3 w = xf_vector(e.w2l, wi);
        const float theta = det_acosf(clampf(w.z, -1.f, 1.f));
        float p = det_atan2f(w.y, w.x);
        const float phi = (p < 0) ? (p + 2 * HPRT_PI) : p;
        const float sinTheta = det_sinf(theta);
        if (sinTheta == 0) return 0;
        return env_pdf_uv(sc, e, phi * HPRT_INV_2PI, theta * HPRT_INV_PI) / (2 * HPRT_PI * HPRT_PI * sinTheta);
    }
    if (GENERIC && l.type == 3) {
        DRay ray; ray.o = offset_ray_origin(ref.p, ref.pErr, ref.n, wi); ray.d = wi; ray.tMax = HPRT_INF;
        float b0, b1, b2, t;
        if (!triangle_may_hit(sc, l.prim, ray, &b0, &b1, &b2, &t)) return 0;
        DevSI isl;
        fill_triangle(sc, (uint32_t)l.prim, b0, b1, b2, wi, &isl);
        const float4 v0 = sc.tris[3 * l.prim], v1 = sc.tris[3 * l.prim + 1], v2 = sc.tris[3 * l.prim + 2];
        float pdf = dist2(ref.p, isl.p) / (absdot(isl.n, -wi) * triangle_area(vec3(v0.x, v0.y, v0.z), vec3(v1.x, v1.y, v1.z), vec3(v2.x, v2.y, v2.z)));
        if (is_inf(pdf)) pdf = 0.f;
        return pdf;
    }
    const DevSphere &s = sc.spheres[l.sphere];
    vec3 pCenter = xf_point(s.o2w, vec3(0, 0, 0));
    if (GENERIC && sphere_ref_inside(s, ref)) {
        DRay ray; ray.o = offset_ray_origin(ref.p, ref.pErr, ref.n, wi); ray.d = wi; ray.tMax = HPRT_INF;
        DevSI isl; float tHit;
        if (!fill_sphere(sc, l.shape, ray, &isl, &tHit)) return 0;
        float area = s.phiMax * s.radius * (s.zMax - s.zMin);
        float pdf = dist2(ref.p, isl.p) / (absdot(isl.n, -wi) * area);
        if (is_inf(pdf)) pdf = 0.f;
        return pdf;
    }
    float sinThetaMax2 = s.radius * s.radius / dist2(ref.p, pCenter);
    float cosThetaMax = sqrtf(sel_max(0.f, 1 - sinThetaMax2));
    return 1 / (2 * HPRT_PI * (1 - cosThetaMax));
}
// Distribution1D::SampleDiscrete (core/sampling.h:86-96) with FindInterval (core/pbrt.h:403-415)
// over LightDistribution::Lookup(p) (integrators/path.cpp:125): the scene's one distribution, or the voxel's
// (SpatialLightDistribution::Lookup, core/lightdistrib.cpp:134-147: Bounds3::Offset, int(offset * nVoxels) clamped)
__device__ __forceinline__ int light_voxel(const DevScene &sc, vec3 p) {
    float o[3] = {p.x - sc.wbMin[0], p.y - sc.wbMin[1], p.z - sc.wbMin[2]};
    int pi[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (sc.wbMax[i] > sc.wbMin[i]) o[i] /= sc.wbMax[i] - sc.wbMin[i];
        int v = (int)(o[i] * sc.voxN[i]);
        pi[i] = v < 0 ? 0 : (v > sc.voxN[i] - 1 ? sc.voxN[i] - 1 : v);
    }
    return (pi[0] * sc.voxN[1] + pi[1]) * sc.voxN[2] + pi[2];
}
// *miss (on-demand voxel tables only): the vertex's voxel has no distribution yet; it has been requested, and the vertex must be
// shaded again after the host has had it computed (nothing may be written for it now)
__device__ __forceinline__ int light_pick(const DevScene &sc, vec3 p, float u, float *pdf, bool *miss) {
    *miss = false;
    if (sc.spatial) {
        int v = light_voxel(sc, p);
        const int n = (int)sc.nLights;
        if (sc.voxSlot) {
            const int row = sc.voxSlot[v];
            if (row < 0) {
                if (row == VOX_EMPTY && atomicCAS(&sc.voxSlot[v], VOX_EMPTY, VOX_REQUESTED) == VOX_EMPTY) sc.voxRequest[atomicAdd(sc.voxRequestCount, 1u)] = (uint32_t)v;
                *miss = true; *pdf = 0.f;
                return 0;
            }
            v = row;
        }
        return dist1d_sample_discrete(sc.voxCdf + (size_t)v * (n + 1), sc.voxFunc + (size_t)v * n, sc.voxFuncInt[v], n, u, pdf);
    }
    return dist1d_sample_discrete(sc.lightCdf, sc.lightFunc, sc.lightFuncInt, (int)sc.nLights, u, pdf);
}
__device__ __forceinline__ float power_heuristic(float fPdf, float gPdf) {   // core/sampling.h:171-174, nf = ng = 1
    float f = 1 * fPdf, g = 1 * gPdf;
    return (f * f) / (f * f + g * g);
}

}  // namespace hprt
