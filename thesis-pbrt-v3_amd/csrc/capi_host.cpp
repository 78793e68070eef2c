// hprt — host half of the C ABI (include/hprt.h): scene front-end, baked scenes,
// BVH build, Halton tables, film resolve, PFM writer.  No HIP calls here; the
// device half lives in capi_device.hip.
#include <algorithm>
#include <new>
#include <stdexcept>
#include <cmath>
#include <limits>
#include <vector>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include "../../include/hprt.h"
#include "bvh_builder.h"
#include "halton_tables.h"
#include "hprt_internal.h"
#include "scene_model.h"
#include "hprt_math.h"
#include "wide_bvh.h"

using namespace hprt;

namespace hprt {
thread_local std::string g_lastError;
int SetError(int code, const std::string &msg) { g_lastError = msg; return code; }
// No exception crosses the C ABI: every exported function that can allocate or parse is a function-try-block that ends here.
int HandleException() {
    try { throw; }
    catch (const std::bad_alloc &) { return SetError(HPRT_E_INVALID, "out of memory (or a size in the input that cannot be real)"); }
    catch (const std::length_error &e) { return SetError(HPRT_E_INVALID, std::string("a size in the input cannot be real: ") + e.what()); }
    catch (const std::exception &e) { return SetError(HPRT_E_INVALID, std::string("unexpected exception: ") + e.what()); }
    catch (...) { return SetError(HPRT_E_INVALID, "unexpected exception"); }
}
}  // namespace hprt

extern "C" {

const char *hprt_last_error(void) { return g_lastError.c_str(); }
const char *hprt_version(void) { return "hprt 0.1 gfx950 (HIP, wave64) host+device"; }

int hprt_model_parse(const char *pbrt_path, const char *const *subst, int n_subst, HprtModel **out) try {
    if (!pbrt_path || !out) return SetError(HPRT_E_INVALID, "hprt_model_parse: null argument");
    std::map<std::string, std::string> sm;
    sm["$acc"] = "\"bvh\"";
    for (int i = 0; i + 1 < 2 * n_subst; i += 2) if (subst && subst[i] && subst[i + 1]) sm[subst[i]] = subst[i + 1];
    HprtModel *m = new HprtModel();
    std::string err;
    if (!ParsePbrtFile(pbrt_path, sm, &m->sc, &err)) { delete m; return SetError(HPRT_E_PARSE, err); }
    if (m->sc.opt.accelerator != "bvh") m->sc.warnings.push_back("Accelerator \"" + m->sc.opt.accelerator + "\" is outside the hot-path scope; \"bvh\" used");
    if (m->sc.opt.integrator != "path") m->sc.warnings.push_back("Integrator \"" + m->sc.opt.integrator + "\" is outside the hot-path scope; \"path\" used");
    if (m->sc.opt.sampler != "halton") m->sc.warnings.push_back("Sampler \"" + m->sc.opt.sampler + "\" is outside the hot-path scope; \"halton\" used");
    *out = m;
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }
int hprt_model_load(const char *baked_path, HprtModel **out) try {
    if (!baked_path || !out) return SetError(HPRT_E_INVALID, "hprt_model_load: null argument");
    HprtModel *m = new HprtModel();
    std::string err;
    if (!LoadBakedScene(baked_path, &m->sc, &err)) { delete m; return SetError(HPRT_E_IO, err); }
    *out = m;
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }
int hprt_model_save(const HprtModel *m, const char *baked_path) try {
    if (!m || !baked_path) return SetError(HPRT_E_INVALID, "hprt_model_save: null argument");
    std::string err;
    if (!SaveBakedScene(m->sc, baked_path, &err)) return SetError(HPRT_E_IO, err);
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }
int hprt_model_save_compact(const HprtModel *m, const char *baked_path) try {
    if (!m || !baked_path) return SetError(HPRT_E_INVALID, "hprt_model_save_compact: null argument");
    std::string err;
    if (!SaveBakedScene(m->sc, baked_path, &err, true)) return SetError(HPRT_E_IO, err);
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }
void hprt_model_destroy(HprtModel *m) { delete m; }

int hprt_model_get_options(const HprtModel *m, HprtRenderOptions *o) try {
    if (!m || !o) return SetError(HPRT_E_INVALID, "hprt_model_get_options: null argument");
    const RenderOptions &p = m->sc.opt;
    o->xres = p.xres; o->yres = p.yres;
    memcpy(o->crop, p.crop, 16);
    memcpy(o->filter_radius, p.filterRadius, 8);
    o->film_scale = p.filmScale; o->max_sample_luminance = p.maxSampleLuminance;
    o->fov = p.fov; o->lens_radius = p.lensRadius; o->focal_distance = p.focalDistance;
    memcpy(o->screen_window, p.screenWindow, 16);
    memcpy(o->camera_to_world, p.cameraToWorld.m, 64); memcpy(o->world_to_camera, p.worldToCamera.m, 64);
    o->spp = p.spp; o->sample_pixel_center = p.samplePixelCenter;
    o->max_depth = p.maxDepth; o->rr_threshold = p.rrThreshold; o->light_strategy = p.lightStrategy;
    o->max_node_prims = p.maxNodePrims; o->isect_cost = p.isectCost; o->trav_cost = p.travCost;
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }
int hprt_model_set_options(HprtModel *m, const HprtRenderOptions *o) try {
    if (!m || !o) return SetError(HPRT_E_INVALID, "hprt_model_set_options: null argument");
    if (o->xres <= 0 || o->yres <= 0 || o->spp <= 0) return SetError(HPRT_E_INVALID, "resolution and spp must be positive");
    RenderOptions &p = m->sc.opt;
    p.xres = o->xres; p.yres = o->yres;
    memcpy(p.crop, o->crop, 16);
    memcpy(p.filterRadius, o->filter_radius, 8);
    p.filmScale = o->film_scale; p.maxSampleLuminance = o->max_sample_luminance;
    p.fov = o->fov; p.lensRadius = o->lens_radius; p.focalDistance = o->focal_distance;
    memcpy(p.screenWindow, o->screen_window, 16);
    memcpy(p.cameraToWorld.m, o->camera_to_world, 64); memcpy(p.worldToCamera.m, o->world_to_camera, 64);
    p.spp = o->spp; p.samplePixelCenter = o->sample_pixel_center;
    p.maxDepth = o->max_depth; p.rrThreshold = o->rr_threshold; p.lightStrategy = o->light_strategy;
    p.maxNodePrims = o->max_node_prims; p.isectCost = o->isect_cost; p.travCost = o->trav_cost;
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }
int hprt_model_counts(const HprtModel *m, uint64_t c[7]) try {
    if (!m || !c) return SetError(HPRT_E_INVALID, "hprt_model_counts: null argument");
    uint64_t tris = 0, spheres = 0;
    for (const ShapeDesc &s : m->sc.shapes) { if (s.kind == kTriangleMesh) tris += s.mesh.nTris(); else ++spheres; }
    c[0] = m->sc.shapes.size(); c[1] = tris + spheres; c[2] = tris; c[3] = spheres;
    c[4] = m->sc.materials.size(); c[5] = m->sc.lights.size(); c[6] = m->sc.textures.size();
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }
int hprt_model_texture_info(const HprtModel *m, uint32_t texture, int32_t info[5], float *max_anisotropy) try {
    if (!m || !info || texture >= m->sc.textures.size()) return SetError(HPRT_E_INVALID, "hprt_model_texture_info: bad argument");
    const TextureDesc &t = m->sc.textures[texture];
    info[0] = (int32_t)t.levels.size(); info[1] = t.trilinear; info[2] = t.wrap; info[3] = t.levels[0].w; info[4] = t.levels[0].h;
    if (max_anisotropy) *max_anisotropy = t.maxAniso;
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }
int hprt_model_texture_level(const HprtModel *m, uint32_t texture, uint32_t level, int32_t wh[2], float *rgb) try {
    if (!m || !wh || texture >= m->sc.textures.size() || level >= m->sc.textures[texture].levels.size())
        return SetError(HPRT_E_INVALID, "hprt_model_texture_level: bad argument");
    const MipLevel &l = m->sc.textures[texture].levels[level];
    wh[0] = l.w; wh[1] = l.h;
    if (rgb) memcpy(rgb, l.rgb.data(), 4 * l.rgb.size());
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }
const char *hprt_model_warnings(const HprtModel *m) {
    static thread_local std::string buf;
    buf.clear();
    if (m) for (const std::string &w : m->sc.warnings) { buf += w; buf += '\n'; }
    return buf.c_str();
}

int hprt_bvh_build(const HprtModel *m, HprtBvh **out) try {
    if (!m || !out) return SetError(HPRT_E_INVALID, "hprt_bvh_build: null argument");
    std::vector<float> lo, hi;
    HprtBvh *b = new HprtBvh();
    b->objects.resize(m->sc.nObjects);
    for (uint32_t o = 0; o < m->sc.nObjects; ++o) {
        ComputeObjectPrimBounds(m->sc, (int)o, &lo, &hi);
        BuildBvh(lo.size() / 3, lo.data(), hi.data(), m->sc.opt.maxNodePrims, m->sc.opt.isectCost, m->sc.opt.travCost, &b->objects[o]);
    }
    ComputePrimBounds(m->sc, b->objects, &lo, &hi);
    BuildBvh(lo.size() / 3, lo.data(), hi.data(), m->sc.opt.maxNodePrims, m->sc.opt.isectCost, m->sc.opt.travCost, &b->tree);
    *out = b;
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }
int hprt_bvh_build_from_bounds(size_t n, const float *bmin, const float *bmax, int maxNodePrims, int isectCost, int travCost,
                               HprtBvh **out) try {
    if (!out || (n && (!bmin || !bmax))) return SetError(HPRT_E_INVALID, "hprt_bvh_build_from_bounds: null argument");
    if (n > 0x7fffffffull) return SetError(HPRT_E_UNSUPPORTED, "more than 2^31 primitives");
    HprtBvh *b = new HprtBvh();
    BuildBvh(n, bmin, bmax, maxNodePrims, isectCost, travCost, &b->tree);
    *out = b;
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }
void hprt_bvh_destroy(HprtBvh *b) { delete b; }
int hprt_bvh_info(const HprtBvh *b, uint32_t info[4], float bounds6[6]) try {
    if (!b || !info) return SetError(HPRT_E_INVALID, "hprt_bvh_info: null argument");
    info[0] = (uint32_t)b->tree.nodes.size(); info[1] = (uint32_t)b->tree.primOrder.size();
    info[2] = (uint32_t)b->tree.nLeaves; info[3] = (uint32_t)b->tree.maxDepth;
    if (bounds6) {
        if (b->tree.nodes.empty()) for (int i = 0; i < 6; ++i) bounds6[i] = 0;
        else { memcpy(bounds6, b->tree.nodes[0].bmin, 12); memcpy(bounds6 + 3, b->tree.nodes[0].bmax, 12); }
    }
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }
int hprt_bvh_copy(const HprtBvh *b, void *nodes32, uint32_t *prim_order) try {
    if (!b) return SetError(HPRT_E_INVALID, "hprt_bvh_copy: null argument");
    if (nodes32 && !b->tree.nodes.empty()) memcpy(nodes32, b->tree.nodes.data(), b->tree.nodes.size() * sizeof(BvhNode));
    if (prim_order && !b->tree.primOrder.empty()) memcpy(prim_order, b->tree.primOrder.data(), b->tree.primOrder.size() * 4);
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }

int hprt_bvh_object_info(const HprtBvh *b, uint32_t object, uint32_t info[4], float bounds6[6]) try {
    if (!b || !info || object >= b->objects.size()) return SetError(HPRT_E_INVALID, "hprt_bvh_object_info: bad argument");
    const BvhTree &t = b->objects[object];
    info[0] = (uint32_t)t.nodes.size(); info[1] = (uint32_t)t.primOrder.size(); info[2] = (uint32_t)t.nLeaves; info[3] = (uint32_t)t.maxDepth;
    if (bounds6) {
        if (t.nodes.empty()) for (int i = 0; i < 6; ++i) bounds6[i] = 0;
        else { memcpy(bounds6, t.nodes[0].bmin, 12); memcpy(bounds6 + 3, t.nodes[0].bmax, 12); }
    }
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }
int hprt_bvh_object_copy(const HprtBvh *b, uint32_t object, void *nodes32, uint32_t *prim_order) try {
    if (!b || object >= b->objects.size()) return SetError(HPRT_E_INVALID, "hprt_bvh_object_copy: bad argument");
    const BvhTree &t = b->objects[object];
    if (nodes32 && !t.nodes.empty()) memcpy(nodes32, t.nodes.data(), t.nodes.size() * sizeof(BvhNode));
    if (prim_order && !t.primOrder.empty()) memcpy(prim_order, t.primOrder.data(), t.primOrder.size() * 4);
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }

int hprt_halton_permutations(uint16_t *out, size_t max_entries, size_t *n_entries) try {
    const std::vector<uint16_t> &p = HaltonPermutations();
    if (n_entries) *n_entries = p.size();
    if (out) memcpy(out, p.data(), 2 * (p.size() < max_entries ? p.size() : max_entries));
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }

// Film::WriteImage, core/film.cpp:266-303 (no splats)
int hprt_film_resolve(const float *xyzw, size_t n, float scale, float *rgb) try {
    if (!xyzw || !rgb) return SetError(HPRT_E_INVALID, "hprt_film_resolve: null argument");
    for (size_t i = 0; i < n; ++i) {
        const float *x = &xyzw[4 * i];
        float *o = &rgb[3 * i];
        o[0] = 3.240479f * x[0] - 1.537150f * x[1] - 0.498535f * x[2];
        o[1] = -0.969256f * x[0] + 1.875991f * x[1] + 0.041556f * x[2];
        o[2] = 0.055648f * x[0] - 0.204043f * x[1] + 1.057311f * x[2];
        float w = x[3];
        if (w != 0) {
            float invWt = 1.0f / w;
            o[0] = sel_max(0.f, o[0] * invWt); o[1] = sel_max(0.f, o[1] * invWt); o[2] = sel_max(0.f, o[2] * invWt);
        }
        // splat term: XYZToRGB(0) = +0 in every channel; v + 1*0 keeps v (and turns -0 into +0)
        o[0] += 0.f; o[1] += 0.f; o[2] += 0.f;
        o[0] *= scale; o[1] *= scale; o[2] *= scale;
    }
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }

// Film::WriteGeneralStatMatrix, core/film.cpp:189-210: "<file minus extension>-<name>.txt", one image row per line
int hprt_write_pixel_stats(const char *prefix, const uint64_t *stats7, int width, int height) try {
    if (!prefix || !stats7 || width <= 0 || height <= 0) return SetError(HPRT_E_INVALID, "hprt_write_pixel_stats: bad argument");
    // the matrices Film::WriteGeneralStats writes (core/film.cpp:170-187), with the index of the value in stats7 (-1: zero for a BVH render)
    static const struct { const char *name; int field; } kMatrices[] = {
        {"primitiveIntersections", 1}, {"primitiveIntersectionsP", 2}, {"kdTreeNodeTraversals", -1}, {"kdTreeNodeTraversalsP", -1},
        {"bspTreeNodeTraversals", -1}, {"bspTreeNodeTraversalsP", -1}, {"leafNodeTraversals", 3}, {"leafNodeTraversalsP", 4}};
    for (const auto &m : kMatrices) {
        const std::string path = std::string(prefix) + "-" + m.name + ".txt";
        FILE *fp = fopen(path.c_str(), "w");
        if (!fp) return SetError(HPRT_E_IO, "cannot create " + path);
        bool ok = true;
        for (int y = 0; y < height && ok; ++y) {
            for (int x = 0; x < width; ++x) {
                const unsigned long long v = m.field < 0 ? 0ull : (unsigned long long)stats7[7 * ((size_t)y * width + x) + m.field];
                ok = ok && fprintf(fp, x ? " %llu" : "%llu", v) > 0;
            }
            ok = ok && fputc('\n', fp) != EOF;
        }
        if (fclose(fp) != 0) ok = false;
        if (!ok) return SetError(HPRT_E_IO, "write error on " + path);
    }
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }

// WritePFM, core/imageio.cpp:437+ : "PF", width height, scale -1 (little endian), rows bottom to top
int hprt_write_pfm(const char *path, const float *rgb, int width, int height) try {
    if (!path || !rgb || width <= 0 || height <= 0) return SetError(HPRT_E_INVALID, "hprt_write_pfm: bad argument");
    FILE *fp = fopen(path, "wb");
    if (!fp) return SetError(HPRT_E_IO, std::string("cannot create ") + path);
    bool ok = fprintf(fp, "PF\n%d %d\n-1.000000\n", width, height) > 0;
    for (int y = height - 1; y >= 0 && ok; --y) ok = fwrite(&rgb[3 * (size_t)y * width], sizeof(float), 3 * (size_t)width, fp) == 3 * (size_t)width;
    if (fclose(fp) != 0) ok = false;
    return ok ? HPRT_OK : SetError(HPRT_E_IO, std::string("write error on ") + path);
} catch (...) { return hprt::HandleException(); }

// Film::MergeFilmTile's accumulation (core/film.cpp:124-131) for cross-tile records on a host copy of the film: per
// destination pixel in ascending source-tile order, exactly as the single-GPU film kernels add them.
int hprt_film_records_merge(float *xyzw, size_t n_pixels, HprtFilmRecord *rec, size_t n) try {
    if (!xyzw || (n && !rec)) return SetError(HPRT_E_INVALID, "hprt_film_records_merge: null argument");
    std::sort(rec, rec + n, [](const HprtFilmRecord &a, const HprtFilmRecord &b) {
        return a.dest_pixel != b.dest_pixel ? a.dest_pixel < b.dest_pixel : a.src_tile < b.src_tile;
    });
    for (size_t i = 0; i < n; ++i) if (rec[i].dest_pixel >= n_pixels) return SetError(HPRT_E_INVALID, "film record outside the film");
    for (size_t i = 0; i < n; ++i) {
        float *px = xyzw + 4 * (size_t)rec[i].dest_pixel;
        px[0] += rec[i].xyz[0]; px[1] += rec[i].xyz[1]; px[2] += rec[i].xyz[2]; px[3] += rec[i].weight;
    }
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }

// Diagnostics hook (not part of include/hprt.h): the reference's unit tests for the host/device-shared math of this path,
// run over the product's own functions (hprt_math.h) on the host:
//   failures[0]  FloatingPoint.NextUpDownFloat   src/tests/fp_tests.cpp:29-47    next_up / next_down
//   failures[1]  Distribution1D.Discrete          src/tests/sampling.cpp:231-282  dist1d_build / dist1d_sample_discrete
__attribute__((visibility("default"))) int hprt_debug_host_selftest(int failures[2]) try {
    if (!failures) return HPRT_E_INVALID;
    const float inf = std::numeric_limits<float>::infinity();
    int f = 0;
    if (!(next_up(-0.f) > 0.f)) ++f;
    if (!(next_down(0.f) < 0.f)) ++f;
    if (!(next_up(inf) == inf)) ++f;
    if (!(next_down(inf) < inf)) ++f;
    if (!(next_down(-inf) == -inf)) ++f;
    if (!(next_up(-inf) > -inf)) ++f;
    // the default-seeded PCG32 stream of the reference's test (core/rng.h:61-62, 86-95)
    uint64_t state = 0x853c49e6748fea9bULL; const uint64_t inc = 0xda3e39cb94b95bdbULL;
    auto next32 = [&]() {
        uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u), rot = (uint32_t)(old >> 59u);
        return (xs >> rot) | (xs << ((~rot + 1u) & 31));
    };
    for (int i = 0; i < 100000; ++i) {
        float v;
        do { v = u2f(next32()); } while (std::isnan(v));
        if (std::isinf(v)) continue;
        if (std::nextafter(v, inf) != next_up(v)) ++f;
        if (std::nextafter(v, -inf) != next_down(v)) ++f;
    }
    failures[0] = f;
    f = 0;
    const float func[4] = {0, 1.f, 0.f, 3.f};
    float cdf[5], funcInt;
    dist1d_build(func, 4, cdf, &funcInt);
    float pdf;
    const float us[7] = {0.f, 0.125f, .24999f, .250001f, 0.625f, 0x1.fffffep-1f, 1.f};
    const int want[7] = {1, 1, 1, 3, 3, 3, 3};
    for (int k = 0; k < 7; ++k) {
        if (dist1d_sample_discrete(cdf, func, funcInt, 4, us[k], &pdf) != want[k]) ++f;
        if (pdf != (want[k] == 1 ? 0.25f : 0.75f)) ++f;
    }
    float u = .25f, uMax = .25f;
    for (int i = 0; i < 20; ++i) { u = next_down(u); uMax = next_up(uMax); }
    for (; u < uMax; u = next_up(u)) {
        int interval = dist1d_sample_discrete(cdf, func, funcInt, 4, u, &pdf);
        if (interval == 3) break;
        if (interval != 1) ++f;
    }
    if (!(u < uMax)) ++f;
    for (; u <= uMax; u = next_up(u))
        if (dist1d_sample_discrete(cdf, func, funcInt, 4, u, &pdf) != 3) ++f;
    failures[1] = f;
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }

// Diagnostics hook (not part of include/hprt.h; tests/test_wide_walk.py): the four-wide records BuildWide (wide_bvh.h) makes of a
// linear node array.  Every leaf gets the "boxed" reference ~firstPrimitive & ~WIDE_LEAF_BOXED (the single-triangle shortcut is
// decided at scene creation, where the vertices are).  out64: cap records of 64 bytes; *n_out: records made; *stack_need: the
// deepest stack a walk can hold.  HPRT_E_UNSUPPORTED when the tree keeps the binary walk (non-finite box, extent beyond the grid).
__attribute__((visibility("default"))) int hprt_debug_wide_build(const void *nodes32, uint32_t n_nodes, void *out64, size_t cap, size_t *n_out, int *stack_need) try {
    if (!nodes32 || !n_out || !stack_need) return SetError(HPRT_E_INVALID, "hprt_debug_wide_build: null argument");
    const BvhNode *nd = (const BvhNode *)nodes32;
    std::vector<int32_t> leafRef(n_nodes, WIDE_NONE);
    for (uint32_t i = 0; i < n_nodes; ++i)
        if ((nd[i].countAxis & 3u) == 3u) leafRef[i] = (int32_t)(~(uint32_t)nd[i].offset & ~WIDE_LEAF_BOXED);
    std::vector<DevWide> wide;
    if (!BuildWide(nd, n_nodes, leafRef.data(), &wide, stack_need)) return SetError(HPRT_E_UNSUPPORTED, "the tree does not fit the four-wide grid");
    *n_out = wide.size();
    if (out64 && cap >= wide.size()) memcpy(out64, wide.data(), wide.size() * sizeof(DevWide));
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }

}  // extern "C"
