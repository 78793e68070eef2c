// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
// C entry points for tests/ (ctypes), __graft_entry__.smoke() and bench.py's
// cpu_baseline leg.  Nothing under thesis-pbrt-v3_amd/ links this.
#include <cstdio>
#include <string>
#include "orc_integrator.h"
#include "orc_selftests.h"

namespace orc { bool g_use_libm = false; }
using namespace orc;

static thread_local std::string g_err;

extern "C" {

const char *orc_last_error() { return g_err.c_str(); }
void orc_set_libm(int on) { g_use_libm = on != 0; }

void *orc_scene_load(const char *path) {
    Renderer *r = new Renderer();
    std::string err;
    if (!LoadScene(path, &r->scene, &err) || !r->Setup(&err)) { g_err = err; delete r; return nullptr; }
    return r;
}
void orc_scene_free(void *h) { delete (Renderer *)h; }

// Override film/sampler parameters after load (crop in pixels fractions as in the scene file)
int orc_set_film(void *h, int xres, int yres, const float *crop4, int spp, int maxDepth) {
    Renderer *r = (Renderer *)h;
    if (xres > 0) r->scene.prm.xres = xres;
    if (yres > 0) r->scene.prm.yres = yres;
    if (crop4) for (int i = 0; i < 4; ++i) r->scene.prm.crop[i] = crop4[i];
    if (spp > 0) r->scene.prm.spp = spp;
    if (maxDepth >= 0) r->scene.prm.maxDepth = maxDepth;
    r->camera.Init(r->scene.prm);
    r->film.Init(r->scene.prm);
    return 0;
}
void orc_film_bounds(void *h, int *out4) {
    Renderer *r = (Renderer *)h;
    out4[0] = r->film.cx0; out4[1] = r->film.cy0; out4[2] = r->film.cx1; out4[3] = r->film.cy1;
}

void orc_bvh_info(void *h, int *nNodes, int *nPrims, int *nLeaves, int *maxDepth, float *bounds6) {
    Renderer *r = (Renderer *)h;
    *nNodes = (int)r->bvh.nodes.size(); *nPrims = (int)r->bvh.primOrder.size();
    *nLeaves = r->bvh.nLeaves; *maxDepth = r->bvh.maxDepth;
    B3 b = r->bvh.WorldBound();
    bounds6[0] = b.pMin.x; bounds6[1] = b.pMin.y; bounds6[2] = b.pMin.z;
    bounds6[3] = b.pMax.x; bounds6[4] = b.pMax.y; bounds6[5] = b.pMax.z;
}
void orc_bvh_copy(void *h, void *nodes, uint32_t *primOrder) {
    Renderer *r = (Renderer *)h;
    memcpy(nodes, r->bvh.nodes.data(), r->bvh.nodes.size() * sizeof(LinearBVHNode));
    memcpy(primOrder, r->bvh.primOrder.data(), r->bvh.primOrder.size() * 4);
}

// counters: [0]nodesFetched [1]nodesFetchedP [2]nodesEntered [3]nodesEnteredP [4]triTests [5]triTestsP
//           [6]triHits [7]triHitsP [8]sphereTests [9]sphereTestsP [10]rays [11]shadowRays [12]cameraRays
// ([4], [6] include Shape::Pdf's tests on triangle emitters as the reference's counters do; [8] counts what the aggregate's traversal tests —
//  the reference has no sphere statistic, the fork's per-ray primitiveIntersections see the traversal's tests only)
static void export_counters(const Counters &c, uint64_t *o) {
    o[0] = c.nodesFetched; o[1] = c.nodesFetchedP; o[2] = c.nodesEntered; o[3] = c.nodesEnteredP;
    o[4] = c.triTests + c.triTestsPdf; o[5] = c.triTestsP; o[6] = c.triHits + c.triHitsPdf; o[7] = c.triHitsP;      // the reference's nTests / nHits: every Triangle::Intersect call
    o[8] = c.sphereTests; o[9] = c.sphereTestsP; o[10] = c.rays; o[11] = c.shadowRays; o[12] = c.cameraRays;
}

// Closest hit for n rays (BVHAccel::Intersect).  prim = index into the ORDERED
// primitive list (-1 on a miss), t = shrunken tMax (unchanged tMax on a miss).
void orc_intersect(void *h, size_t n, const float *o, const float *d, const float *tmax, float *t, int32_t *prim,
                   float *bary, uint64_t *counters13) {
    Renderer *r = (Renderer *)h;
    Counters ctr;
    for (size_t i = 0; i < n; ++i) {
        Ray ray(V3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), V3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmax[i]);
        SurfaceInteraction si; int ordered = -1;
        bool hit = r->bvh.Intersect(ray, &si, ctr, &ordered);
        t[i] = ray.tMax; prim[i] = hit ? ordered : -1;
        bary[3 * i] = hit ? si.b0 : 0; bary[3 * i + 1] = hit ? si.b1 : 0; bary[3 * i + 2] = hit ? si.b2 : 0;
    }
    if (counters13) export_counters(ctr, counters13);
}
// Same with the instance each hit went through (-1: none).  prim numbers the ordered primitives of
// all aggregates: top level first, then object 0, 1, ...
void orc_intersect_inst(void *h, size_t n, const float *o, const float *d, const float *tmax, float *t, int32_t *prim,
                        int32_t *inst, float *bary, uint64_t *counters13) {
    Renderer *r = (Renderer *)h;
    Counters ctr;
    for (size_t i = 0; i < n; ++i) {
        Ray ray(V3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), V3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmax[i]);
        SurfaceInteraction si; int ordered = -1;
        bool hit = r->bvh.Intersect(ray, &si, ctr, &ordered);
        t[i] = ray.tMax; prim[i] = hit ? ordered : -1; inst[i] = hit ? si.inst : -1;
        bary[3 * i] = hit ? si.b0 : 0; bary[3 * i + 1] = hit ? si.b1 : 0; bary[3 * i + 2] = hit ? si.b2 : 0;
    }
    if (counters13) export_counters(ctr, counters13);
}
int orc_object_count(void *h) { return (int)((Renderer *)h)->objectBvh.size(); }
void orc_object_bvh_info(void *h, int object, int *nNodes, int *nPrims) {
    Renderer *r = (Renderer *)h;
    *nNodes = (int)r->objectBvh[object].nodes.size(); *nPrims = (int)r->objectBvh[object].primOrder.size();
}
void orc_object_bvh_copy(void *h, int object, void *nodes, uint32_t *primOrder) {
    Renderer *r = (Renderer *)h;
    memcpy(nodes, r->objectBvh[object].nodes.data(), r->objectBvh[object].nodes.size() * sizeof(LinearBVHNode));
    memcpy(primOrder, r->objectBvh[object].primOrder.data(), r->objectBvh[object].primOrder.size() * 4);
}
// Per-ray work of the closest-hit walk: nodes fetched and primitive tests (for SIMT-efficiency studies)
void orc_intersect_work(void *h, size_t n, const float *o, const float *d, const float *tmax, uint32_t *nodes, uint32_t *prims) {
    Renderer *r = (Renderer *)h;
    for (size_t i = 0; i < n; ++i) {
        Counters ctr;
        Ray ray(V3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), V3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmax[i]);
        SurfaceInteraction si;
        r->bvh.Intersect(ray, &si, ctr);
        nodes[i] = (uint32_t)ctr.nodesFetched; prims[i] = (uint32_t)(ctr.triTests + ctr.sphereTests);
    }
}
// Per-ray fetch traces, concatenated; offsets[n+1] (study aid for SIMT scheduling)
size_t orc_intersect_trace(void *h, size_t n, const float *o, const float *d, const float *tmax, uint8_t *trace, size_t cap, uint64_t *offsets) {
    Renderer *r = (Renderer *)h;
    std::vector<uint8_t> tr;
    r->bvh.fetchTrace = &tr;
    size_t pos = 0;
    for (size_t i = 0; i < n; ++i) {
        tr.clear();
        Counters ctr; SurfaceInteraction si;
        Ray ray(V3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), V3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmax[i]);
        r->bvh.Intersect(ray, &si, ctr);
        offsets[i] = pos;
        for (uint8_t v : tr) { if (pos < cap) trace[pos] = v; ++pos; }
    }
    offsets[n] = pos;
    r->bvh.fetchTrace = nullptr;
    return pos;
}
// Same, also returning the SurfaceInteraction fill (p, pError, n, shading.n, shading.dpdu)
void orc_intersect_full(void *h, size_t n, const float *o, const float *d, const float *tmax, float *t,
                        int32_t *prim, float *si15) {
    Renderer *r = (Renderer *)h;
    Counters ctr;
    for (size_t i = 0; i < n; ++i) {
        Ray ray(V3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), V3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmax[i]);
        SurfaceInteraction si; int ordered = -1;
        bool hit = r->bvh.Intersect(ray, &si, ctr, &ordered);
        t[i] = ray.tMax; prim[i] = hit ? ordered : -1;
        float *q = si15 + 15 * i;
        if (hit) {
            V3 v[5] = {si.p, si.pError, si.n, si.shading.n, si.shading.dpdu};
            for (int k = 0; k < 5; ++k) { q[3 * k] = v[k].x; q[3 * k + 1] = v[k].y; q[3 * k + 2] = v[k].z; }
        } else for (int k = 0; k < 15; ++k) q[k] = 0;
    }
}
void orc_occluded(void *h, size_t n, const float *o, const float *d, const float *tmax, uint8_t *occ,
                  uint64_t *counters13) {
    Renderer *r = (Renderer *)h;
    Counters ctr;
    for (size_t i = 0; i < n; ++i) {
        Ray ray(V3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), V3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmax[i]);
        occ[i] = r->bvh.IntersectP(ray, ctr) ? 1 : 0;
    }
    if (counters13) export_counters(ctr, counters13);
}

// Sampler known answers
void orc_pcg32(int n, uint32_t *out) { RNG rng; for (int i = 0; i < n; ++i) out[i] = rng.UniformUInt32(); }
int orc_perm_table(uint16_t *out, int maxEntries) {
    const std::vector<uint16_t> &p = HaltonSampler::Permutations();
    int n = (int)p.size();
    for (int i = 0; i < n && i < maxEntries; ++i) out[i] = p[i];
    return n;
}
float orc_radical_inverse(int baseIndex, uint64_t a) { return RadicalInverse(baseIndex, a); }
float orc_scrambled_radical_inverse(int baseIndex, uint64_t a) {
    return ScrambledRadicalInverse(baseIndex, a, &HaltonSampler::Permutations()[PrimeSums()[baseIndex]]);
}
// dims [dim0, dim0+nd) of sample `sampleNum` at pixel (px,py) for a sampler over [sx0,sx1)x[sy0,sy1)
int64_t orc_halton(int sx0, int sy0, int sx1, int sy1, int px, int py, int64_t sampleNum, int dim0, int nd,
                   float *out) {
    HaltonSampler s(1 << 30, sx0, sy0, sx1, sy1);
    s.StartPixel(px, py);
    s.SetSampleNumber(sampleNum);
    for (int i = 0; i < nd; ++i) out[i] = s.SampleDimension(s.intervalSampleIndex, dim0 + i);
    return s.intervalSampleIndex;
}

// Camera rays for (px,py,sampleNum) triples
void orc_camera_rays(void *h, size_t n, const int *px, const int *py, const int64_t *sn, float *o, float *d) {
    Renderer *r = (Renderer *)h;
    int sx0, sy0, sx1, sy1; r->film.GetSampleBounds(&sx0, &sy0, &sx1, &sy1);
    HaltonSampler s(1 << 30, sx0, sy0, sx1, sy1, r->scene.prm.samplePixelCenter != 0);
    for (size_t i = 0; i < n; ++i) {
        s.StartPixel(px[i], py[i]); s.SetSampleNumber(sn[i]);
        P2 u = s.Get2D(); s.Get1D(); P2 pl = s.Get2D();
        Ray ray; r->camera.GenerateRay(P2((Float)px[i] + u.x, (Float)py[i] + u.y), pl, &ray);
        o[3 * i] = ray.o.x; o[3 * i + 1] = ray.o.y; o[3 * i + 2] = ray.o.z;
        d[3 * i] = ray.d.x; d[3 * i + 1] = ray.d.y; d[3 * i + 2] = ray.d.z;
    }
}
// Radiance of individual camera samples (after the NaN/negative/inf guards)
void orc_sample_radiance(void *h, size_t n, const int *px, const int *py, const int64_t *sn, float *L3) {
    Renderer *r = (Renderer *)h;
    int sx0, sy0, sx1, sy1; r->film.GetSampleBounds(&sx0, &sy0, &sx1, &sy1);
    HaltonSampler s(1 << 30, sx0, sy0, sx1, sy1, r->scene.prm.samplePixelCenter != 0);
    // any sample index may be asked for, but the samples belong to the frame of the scene's spp: the camera ray differentials
    // are scaled by 1/sqrt(samplesPerPixel) (core/integrator.cpp:288-289; only image textures see them)
    s.samplesPerPixel = r->scene.prm.spp;
    Counters ctr;
    for (size_t i = 0; i < n; ++i) {
        s.StartPixel(px[i], py[i]); s.SetSampleNumber(sn[i]);
        P2 pf; Float w;
        Spec L = r->RenderSample(s, px[i], py[i], &pf, &w, ctr);
        L3[3 * i] = L.c[0]; L3[3 * i + 1] = L.c[1]; L3[3 * i + 2] = L.c[2];
    }
}

// Full render.  rgb: 3*W*H floats, top row first (W,H = cropped pixel bounds).
int orc_render(void *h, int spp, int nThreads, float *rgb, uint64_t *counters13, double *seconds) {
    Renderer *r = (Renderer *)h;
    r->nThreads = nThreads > 0 ? nThreads : (int)std::thread::hardware_concurrency();
    r->Render(spp);
    if (rgb) {
        std::vector<Float> out;
        FilmToRGB(&r->film, &out);
        memcpy(rgb, out.data(), out.size() * sizeof(Float));
    }
    if (counters13) export_counters(r->total, counters13);
    if (seconds) *seconds = r->renderSeconds;
    return r->nThreads;
}
// Render only tiles tile_begin, tile_begin+stride, ... (one rank of a tile-sharded job); film state only
int orc_render_tiles(void *h, int spp, int nThreads, int tileBegin, int tileStride, uint64_t *counters13) {
    Renderer *r = (Renderer *)h;
    r->nThreads = nThreads > 0 ? nThreads : (int)std::thread::hardware_concurrency();
    r->Render(spp, tileBegin, tileStride > 0 ? tileStride : 1);
    if (counters13) export_counters(r->total, counters13);
    return r->nThreads;
}
// Per-pixel GeneralStats of the last render: 7 uint64 per pixel (see Film::stats)
void orc_pixel_stats(void *h, uint64_t *out7) {
    Renderer *r = (Renderer *)h;
    memcpy(out7, r->film.stats.data(), r->film.stats.size() * 7 * sizeof(uint64_t));
}
// Raw film state (xyz + weight per pixel) for exact film comparisons
void orc_film_raw(void *h, float *xyzw) {
    Renderer *r = (Renderer *)h;
    for (size_t i = 0; i < r->film.pixels.size(); ++i) {
        xyzw[4 * i] = r->film.pixels[i].xyz[0]; xyzw[4 * i + 1] = r->film.pixels[i].xyz[1];
        xyzw[4 * i + 2] = r->film.pixels[i].xyz[2]; xyzw[4 * i + 3] = r->film.pixels[i].filterWeightSum;
    }
}

float orc_scrambled_radical_inverse_perm(int baseIndex, uint64_t a, const uint16_t *perm) {
    return ScrambledRadicalInverse(baseIndex, a, perm);
}
int orc_prime(int i) { return Primes()[i]; }
// single-triangle Triangle::Intersect (mesh without N/uv/S)
int orc_triangle_intersect(const float *p9, const float *o, const float *d, float tmax, float *tHit) {
    Mesh m; m.nTris = 1; m.nVerts = 3; m.hasN = m.hasUV = m.hasS = false;
    m.idx = {0, 1, 2};
    for (int i = 0; i < 3; ++i) m.p.push_back(V3(p9[3 * i], p9[3 * i + 1], p9[3 * i + 2]));
    TriRef tr{&m, m.idx.data(), false};
    Ray ray(V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), tmax);
    SurfaceInteraction si; Counters c; Float t = 0;
    bool hit = TriangleIntersect(tr, ray, &t, &si, c);
    if (tHit) *tHit = t;
    return hit ? 1 : 0;
}
// the reference's own property tests, restated over the oracle (orc_selftests.h)
int orc_selftest_watertight(int nRays) { return SelfTestWatertight(nRays); }
int orc_selftest_reintersect(int nTriangles, int nRaysPerTriangle, int *nTested) { return SelfTestReintersect(nTriangles, nRaysPerTriangle, nTested); }
int orc_selftest_radical_inverse() { return SelfTestRadicalInverse(); }
int orc_selftest_sphere_pretest(int nRays, int *nFull, int *nMaybe) { return SelfTestSpherePretest(nRays, nFull, nMaybe); }
int orc_selftest_scrambled_radical_inverse() { return SelfTestScrambledRadicalInverse(); }
int orc_selftest_sphere_reintersect(int nSpheres, int nRays, int *nTested) { return SelfTestSphereReintersect(nSpheres, nRays, nTested); }
int orc_selftest_next_float() { return SelfTestNextFloat(); }
int orc_selftest_efloat(int iters) { return SelfTestEFloat(iters); }
int orc_selftest_bsdf_sampling(int which, double *minPval) { return SelfTestBSDFSampling(which, minPval); }
int orc_selftest_distribution1d() { return SelfTestDistribution1D(); }
// inputs of the reference's property tests for the batched device queries (tests/test_gpu_reference_unit_tests.py)
void orc_watertight_case(int nIter, float *P, int *idx, float *o, float *d, float *tBrute) { WatertightCase(nIter, P, idx, o, d, tBrute); }
// params: radius, zMin, zMax, thetaMin, thetaMax, phiMax (as Sphere holds them); first: o, d, tMax; returns 1 if the first ray hits
int orc_sphere_reintersect_case(int seed, int partial, int nRays, float *params, float *first, float *tFirst, float *rays, int *failures) {
    Sphere s; Ray r; Float t = 0; int f = 0;
    const bool hit = SphereReintersectCase(seed, partial != 0, nRays, &s, &r, &t, rays, &f);
    params[0] = s.radius; params[1] = s.zMin; params[2] = s.zMax; params[3] = s.thetaMin; params[4] = s.thetaMax; params[5] = s.phiMax;
    first[0] = r.o.x; first[1] = r.o.y; first[2] = r.o.z; first[3] = r.d.x; first[4] = r.d.y; first[5] = r.d.z; first[6] = r.tMax;
    *tFirst = t; *failures = f;
    return hit ? 1 : 0;
}

// detmath probes
float orc_det_sinf(float x) { return det::sinf_glibc(x); }
float orc_det_cosf(float x) { return det::cosf_glibc(x); }
// Number of floats with bit patterns first, first+stride, ... (count of them, |x| < 120 only) on which the restated sinf / cosf
// differ from the libm this process is linked with; out[0] = sinf mismatches, out[1] = cosf mismatches.
void orc_sincosf_vs_libm(uint32_t first, uint32_t stride, uint64_t count, uint64_t out[2]) {
    uint64_t bs = 0, bc = 0;
    uint32_t u = first;
    for (uint64_t i = 0; i < count; ++i, u += stride) {
        float y; memcpy(&y, &u, 4);
        if (!(std::fabs(y) < 120.f)) continue;
        volatile float yy = y;
        float a = sinf(yy), b = det::sinf_glibc(y);
        if (memcmp(&a, &b, 4)) ++bs;
        a = cosf(yy); b = det::cosf_glibc(y);
        if (memcmp(&a, &b, 4)) ++bc;
    }
    out[0] = bs; out[1] = bc;
}
double orc_det_sin(double x) { return det::sin_d(x); }
double orc_det_cos(double x) { return det::cos_d(x); }
double orc_det_sin_glibc(double x) { return det::sin_glibc_d(x); }
double orc_det_cos_glibc(double x) { return det::cos_glibc_d(x); }
// Array form of the probes above for the GPU-side comparison (tests/test_gpu_math.py): same fn numbering and outputs as the
// product's hprt_debug_device_math; libm != 0 evaluates this machine's libm instead of the restatements.
void orc_math_eval(int fn, int libm, const float *x, const float *y, uint64_t n, double *o0, double *o1) {
    for (uint64_t i = 0; i < n; ++i) {
        double a = 0, b = 0;
        volatile float xv = x[i], yv = y[i];
        if (fn == 0) { a = libm ? (double)sinf(xv) : (double)det::sinf_glibc(x[i]); b = libm ? (double)cosf(xv) : (double)det::cosf_glibc(x[i]); }
        else if (fn == 1) a = libm ? (double)acosf(xv) : (double)det::acosf_glibc(x[i]);
        else if (fn == 2) a = libm ? (double)atan2f(yv, xv) : (double)det::atan2f_glibc(y[i], x[i]);
        else if (fn == 3) a = libm ? (double)logf(xv) : (double)det::logf_glibc(x[i]);
        else { volatile double xd = (double)x[i]; a = libm ? ::sin(xd) : det::sin_glibc_d((double)x[i]); b = libm ? ::cos(xd) : det::cos_glibc_d((double)x[i]); }
        o0[i] = a; o1[i] = b;
    }
}
// The restated double sin / cos against this process's libm on the float arguments with bit patterns first, first+stride, ...
// (count of them) inside [0, 2 pi) — the domain of the one call site, core/microfacet.cpp:243-245; out[0] = sin, out[1] = cos mismatches
void orc_sincos_d_vs_libm(uint32_t first, uint32_t stride, uint64_t count, uint64_t out[2]) {
    uint64_t bs = 0, bc = 0;
    uint32_t u = first;
    for (uint64_t i = 0; i < count; ++i, u += stride) {
        if (u > 0x40c90fdau) break;                     // (float)(6.28318530718 * 0x1.fffffep-1)
        float y; memcpy(&y, &u, 4);
        volatile double yy = (double)y;
        double a = ::sin(yy), b = det::sin_glibc_d((double)y);
        if (memcmp(&a, &b, 8)) ++bs;
        a = ::cos(yy); b = det::cos_glibc_d((double)y);
        if (memcmp(&a, &b, 8)) ++bc;
    }
    out[0] = bs; out[1] = bc;
}
float orc_det_atan2f(float y, float x) { return det::atan2f_glibc(y, x); }
float orc_det_acosf(float x) { return det::acosf_glibc(x); }
// logf mismatches against libm over positive finite float bit patterns first, first+stride, ...
uint64_t orc_logf_vs_libm(uint32_t first, uint32_t stride, uint64_t count) {
    uint64_t bad = 0; uint32_t u = first;
    for (uint64_t i = 0; i < count; ++i, u += stride) {
        if (u == 0 || u >= 0x7f800000u) continue;
        float x; memcpy(&x, &u, 4);
        volatile float xx = x;
        float a = logf(xx), b = det::logf_glibc(x);
        if (memcmp(&a, &b, 4)) ++bad;
    }
    return bad;
}
// as orc_sincosf_vs_libm: out[0] = acosf mismatches (|x| <= 1), out[1] = atanf mismatches, out[2] = atan2f(y, x) mismatches with x
// taken from a second pattern sequence
void orc_atan_acos_vs_libm(uint32_t first, uint32_t stride, uint64_t count, uint64_t out[3]) {
    uint64_t ba = 0, bt = 0, b2 = 0;
    uint32_t u = first, v = first * 2654435761u + 12345u;
    for (uint64_t i = 0; i < count; ++i, u += stride, v = v * 1664525u + 1013904223u) {
        float y, x; memcpy(&y, &u, 4); memcpy(&x, &v, 4);
        if (y != y) continue;
        volatile float yy = y, xx = x;
        float a, b;
        if (std::fabs(y) <= 1.f) { a = acosf(yy); b = det::acosf_glibc(y); if (memcmp(&a, &b, 4)) ++ba; }
        a = atanf(yy); b = det::atanf_glibc(y); if (memcmp(&a, &b, 4)) ++bt;
        if (x == x && !std::isinf(x) && !std::isinf(y)) { a = atan2f(yy, xx); b = det::atan2f_glibc(y, x); if (memcmp(&a, &b, 4)) ++b2; }
    }
    out[0] = ba; out[1] = bt; out[2] = b2;
}

}  // extern "C"
