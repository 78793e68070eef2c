// hprt — device half of the C ABI (include/hprt.h): HBM scene upload, batched
// Intersect/IntersectP, the wavefront Render loop and the film.  Host-side
// orchestration only; the arithmetic lives in device/*.h and device/kernels.hip.
// There is no CPU fallback: every entry point fails with HPRT_E_NO_DEVICE when no
// HIP device is usable.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "../../include/hprt.h"
#include "device/kernels.h"
#include "halton_tables.h"
#include "host_transform.h"
#include "hprt_internal.h"
#include "device_state.h"
#include "wide_bvh.h"

using namespace hprt;


namespace {

int CheckDevice(int device, int *chosen) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return SetError(HPRT_E_NO_DEVICE, "no HIP device available (hprt has no CPU fallback)");
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= n) return SetError(HPRT_E_INVALID, "device ordinal out of range");
    if (hipSetDevice(device) != hipSuccess) return SetError(HPRT_E_DEVICE, "hipSetDevice failed");
    *chosen = device;
    return HPRT_OK;
}

// TrowbridgeReitzDistribution::RoughnessToAlpha, core/microfacet.h:123-128 (host libm logf,
// as in the reference; constant textures make it a per-material constant)
float RoughnessToAlpha(float roughness) {
    roughness = sel_max(roughness, (float)1e-3);
    float x = std::log(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}

// Would Triangle::Intersect reject every hit on this triangle as "bogus"
// (shapes/triangle.cpp:300-316)?  Depends on the triangle only, so decided here.
bool TriangleIsBogus(vec3 p0, vec3 p1, vec3 p2, const float *uv0, const float *uv1, const float *uv2) {
    float u0x = 0, u0y = 0, u1x = 1, u1y = 0, u2x = 1, u2y = 1;
    if (uv0) { u0x = uv0[0]; u0y = uv0[1]; u1x = uv1[0]; u1y = uv1[1]; u2x = uv2[0]; u2y = uv2[1]; }
    float duv02x = u0x - u2x, duv02y = u0y - u2y, duv12x = u1x - u2x, duv12y = u1y - u2y;
    vec3 dp02 = p0 - p2, dp12 = p1 - p2;
    float determinant = duv02x * duv12y - duv02y * duv12x;
    bool degenerateUV = std::fabs(determinant) < 1e-8;
    vec3 dpdu, dpdv;
    if (!degenerateUV) {
        float invdet = 1 / determinant;
        dpdu = (duv12y * dp02 - duv02y * dp12) * invdet;
        dpdv = (-duv12x * dp02 + duv02x * dp12) * invdet;
    }
    if (degenerateUV || length2(cross(dpdu, dpdv)) == 0) {
        vec3 ng = cross(p2 - p0, p1 - p0);
        if (length2(ng) == 0) return true;
    }
    return false;
}

struct PlaneAllocator {
    char *base; size_t off = 0, cap;
    template <typename T> T *take(size_t n) { off = (off + 255) & ~(size_t)255; T *p = (T *)(base + off); off += n * sizeof(T); return p; }
};
// Device view of one batch's path data (device/kernels.h): two path streams used alternately
// (a bounce reads one and writes the other), the hits of the stream being read, the per-vertex
// streams and the finished paths' radiance by path id.
struct Workspace {
    PathStream path[2];
    HitStream hit;
    VertexStreams vs;
    float4 *Lfinal;
};
// 16-byte words per stream index: 2 x (ray a,b + beta + L) + hit a + shadow a,b + mis a,b + misHit a
// + pendLight/Mis/Beta + Lfinal; plus b2 + instance of the path hit and of the MIS hit (8 B each), occluded (1 B) and alignment slack
const size_t kPlaneBytesPerSlot = 16 * (2 * 4 + 1 + 2 + 2 + 1 + 3 + 1) + 8 + 8 + 1;
size_t PlaneBytes(size_t n) { return n * kPlaneBytesPerSlot + 32 * 256; }

void CarvePlanes(char *base, size_t n, Workspace *w) {
    PlaneAllocator a{base, 0, 0};
    auto rays = [&](RayStream &r) { r.a = a.take<float4>(n); r.b = a.take<float4>(n); };
    for (int k = 0; k < 2; ++k) { rays(w->path[k].ray); w->path[k].beta = a.take<float4>(n); w->path[k].L = a.take<float4>(n); }
    w->hit.a = a.take<float4>(n); w->hit.b = a.take<float2>(n);
    rays(w->vs.shadow); w->vs.occluded = a.take<uint8_t>(n);
    rays(w->vs.mis); w->vs.misHit.a = a.take<float4>(n); w->vs.misHit.b = a.take<float2>(n);      // (b2: the emitter's shading normal at a triangle light)
    w->vs.pendLight = a.take<float4>(n); w->vs.pendMis = a.take<float4>(n); w->vs.pendBeta = a.take<float4>(n);
    w->Lfinal = a.take<float4>(n);
}

struct EventTimer {
    std::vector<hipEvent_t> pool; size_t used = 0;
    ~EventTimer() { for (hipEvent_t e : pool) (void)hipEventDestroy(e); }
    hipEvent_t get() {
        if (used == pool.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; pool.push_back(e); }
        return pool[used++];
    }
    void reset() { used = 0; }
};

// Camera matrices: ProjectiveCamera ctor, core/camera.h:84-115
void MakeCamera(const HprtRenderOptions &o, DevCamera *cam) {
    Xform camToScreen = xf_perspective(o.fov, 1e-2f, 1000.f);
    const float *sw = o.screen_window;
    Xform screenToRaster = xf_scale((float)o.xres, (float)o.yres, 1) * xf_scale(1 / (sw[1] - sw[0]), 1 / (sw[2] - sw[3]), 1) *
                           xf_translate(vec3(-sw[0], -sw[3], 0));
    Xform rasterToScreen = screenToRaster.inverse();
    Xform rasterToCamera = camToScreen.inverse() * rasterToScreen;
    cam->rasterToCamera = rasterToCamera.m;
    memcpy(cam->cameraToWorld.m, o.camera_to_world, 64);
    cam->lensRadius = o.lens_radius; cam->focalDistance = o.focal_distance;
    // cameras/perspective.cpp:55-58
    cam->dxCamera = xf_point(cam->rasterToCamera, vec3(1, 0, 0)) - xf_point(cam->rasterToCamera, vec3(0, 0, 0));
    cam->dyCamera = xf_point(cam->rasterToCamera, vec3(0, 1, 0)) - xf_point(cam->rasterToCamera, vec3(0, 0, 0));
}

// RadicalInverse(0..4, i), i < 128 (core/lowdiscrepancy.cpp:2478-2488, 389-403): the 3D point and the 2D light sample of
// SpatialLightDistribution::ComputeDistribution's 128 samples per voxel, as [5][128]
void VoxelSamplePoints(float *out) {
    const int bases[5] = {2, 3, 5, 7, 11};
    for (int b = 0; b < 5; ++b)
        for (uint32_t i = 0; i < 128; ++i) {
            float v;
            if (b == 0) {
                uint32_t n = i;      // ReverseBits32(a) * 0x1p-32, evaluated in double
                n = (n << 16) | (n >> 16); n = ((n & 0x00ff00ffu) << 8) | ((n & 0xff00ff00u) >> 8); n = ((n & 0x0f0f0f0fu) << 4) | ((n & 0xf0f0f0f0u) >> 4);
                n = ((n & 0x33333333u) << 2) | ((n & 0xccccccccu) >> 2); n = ((n & 0x55555555u) << 1) | ((n & 0xaaaaaaaau) >> 1);
                v = (float)((double)n * 0x1p-32);
            } else {
                const int base = bases[b];
                const float invBase = (float)1 / (float)base;
                uint64_t reversedDigits = 0, a = i;
                float invBaseN = 1;
                while (a) { uint64_t next = a / base, digit = a - next * base; reversedDigits = reversedDigits * base + digit; invBaseN *= invBase; a = next; }
                v = sel_min((float)reversedDigits * invBaseN, HPRT_ONE_MINUS_EPS);
            }
            out[128 * b + i] = v;
        }
}

// A path vertex consumes up to 8 sampler dimensions after the camera sample's 5; the reference's
// Halton sampler aborts at dimension PrimeTableSize = 1000 (core/lowdiscrepancy.h:52, lowdiscrepancy.cpp:4558), so depths
// whose paths could get there are refused instead of sampled from a dimension the reference does not have.
int CheckDepth(int maxDepth) {
    if (5 + 8 * ((int64_t)maxDepth + 1) > 1000)
        return SetError(HPRT_E_UNSUPPORTED, "maxdepth above 123: a path could pass the 1,000 sampler dimensions of the reference's Halton tables");
    return HPRT_OK;
}

struct FrameSetup {
    FilmGeom fg; int ntx, nty; HaltonLayout hal; RenderParams rp;
    std::vector<uint32_t> pixelXY; std::vector<uint64_t> pixelOffset;
    std::vector<int32_t> localIndex;   // cropped-film index -> local pixel index or -1
    int W, H;
    // HaltonSampler::GetIndexForSample's per-pixel offset (samplers/halton.cpp:101-118) depends on the pixel
    // coordinates modulo kMaxResolution only: offset = (offX[x mod 128] + offY[y mod 128]) mod sampleStride
    uint64_t offX[128], offY[128];
};
int SetupFrame(const HprtRenderOptions &o, FrameSetup *f) {
    if (o.xres <= 0 || o.yres <= 0 || o.xres > 32768 || o.yres > 32768) return SetError(HPRT_E_INVALID, "film resolution out of range");
    if (o.filter_radius[0] != 0.5f || o.filter_radius[1] != 0.5f)
        return SetError(HPRT_E_UNSUPPORTED, "only the box filter with radius 0.5 is in the hot-path scope (SURVEY.md §2)");
    FilmGeom &fg = f->fg;
    fg.cx0 = (int)std::ceil(o.xres * o.crop[0]); fg.cx1 = (int)std::ceil(o.xres * o.crop[1]);   // core/film.cpp:56-60
    fg.cy0 = (int)std::ceil(o.yres * o.crop[2]); fg.cy1 = (int)std::ceil(o.yres * o.crop[3]);
    fg.rx = o.filter_radius[0]; fg.ry = o.filter_radius[1];
    fg.sx0 = (int)std::floor((float)fg.cx0 + 0.5f - fg.rx); fg.sy0 = (int)std::floor((float)fg.cy0 + 0.5f - fg.ry);   // core/film.cpp:81-87
    fg.sx1 = (int)std::ceil((float)fg.cx1 - 0.5f + fg.rx); fg.sy1 = (int)std::ceil((float)fg.cy1 - 0.5f + fg.ry);
    fg.maxSampleLuminance = o.max_sample_luminance;
    f->W = fg.cx1 - fg.cx0; f->H = fg.cy1 - fg.cy0;
    if (f->W <= 0 || f->H <= 0) return SetError(HPRT_E_INVALID, "empty crop window");
    f->ntx = (fg.sx1 - fg.sx0 + 15) / 16; f->nty = (fg.sy1 - fg.sy0 + 15) / 16;                   // core/integrator.cpp:237-239
    f->hal = MakeHaltonLayout(fg.sx1 - fg.sx0, fg.sy1 - fg.sy0);
    // the two addends of the offset (each already reduced), from the layout's own function on (m, 0) and (0, m)
    const uint64_t base = (uint64_t)HaltonPixelOffset(f->hal, 0, 0), stride = (uint64_t)std::max(1, f->hal.sampleStride);
    for (int m = 0; m < 128; ++m) {
        f->offX[m] = (uint64_t)HaltonPixelOffset(f->hal, m, 0);
        f->offY[m] = ((uint64_t)HaltonPixelOffset(f->hal, 0, m) + stride - base) % stride;
    }
    return HPRT_OK;
}
void AddTilePixels(FrameSetup *f, int tile) {
    const FilmGeom &fg = f->fg;
    int tx = tile % f->ntx, ty = tile / f->ntx;
    int x0 = fg.sx0 + tx * 16, x1 = std::min(x0 + 16, fg.sx1), y0 = fg.sy0 + ty * 16, y1 = std::min(y0 + 16, fg.sy1);
    const uint64_t stride = (uint64_t)std::max(1, f->hal.sampleStride);
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
            f->localIndex[(size_t)(y - fg.cy0) * f->W + (x - fg.cx0)] = (int32_t)f->pixelXY.size();
            f->pixelXY.push_back((uint32_t)x | ((uint32_t)y << 16));
            f->pixelOffset.push_back(f->hal.sampleStride > 1 ? (f->offX[((x % 128) + 128) % 128] + f->offY[((y % 128) + 128) % 128]) % stride : 0ull);
        }
}

}  // namespace

extern "C" {

// MIPMap<RGBSpectrum>::Lookup(st, width) on the host, over an HprtTextureDesc (core/mipmap.h:203-260; repeat wrap): the same
// float operations as the device's mip_triangle / the oracle's MipLookupWidth.  Used once per infinite light for the scalar image
// of its Distribution2D (lights/infinite.cpp:66-85) and for Power() (:87-91).
namespace {
inline int HostModI(int a, int b) { const int r = a - (a / b) * b; return r < 0 ? r + b : r; }
inline rgb HostMipTexel(const HprtTextureDesc &tx, int level, int s, int t) {
    const HprtTextureLevel &l = tx.levels[level];
    s = HostModI(s, l.w); t = HostModI(t, l.h);
    const float *p = l.rgb + 3 * ((size_t)t * (size_t)l.w + (size_t)s);
    return rgb(p[0], p[1], p[2]);
}
inline rgb HostMipTriangle(const HprtTextureDesc &tx, int level, float su, float sv) {
    level = level < 0 ? 0 : (level > (int)tx.n_levels - 1 ? (int)tx.n_levels - 1 : level);
    const HprtTextureLevel &l = tx.levels[level];
    const float s = su * l.w - 0.5f, t = sv * l.h - 0.5f;
    const int s0 = (int)std::floor(s), t0 = (int)std::floor(t);
    const float ds = s - s0, dt = t - t0;
    return (1 - ds) * (1 - dt) * HostMipTexel(tx, level, s0, t0) + (1 - ds) * dt * HostMipTexel(tx, level, s0, t0 + 1) +
           ds * (1 - dt) * HostMipTexel(tx, level, s0 + 1, t0) + ds * dt * HostMipTexel(tx, level, s0 + 1, t0 + 1);
}
inline rgb HostMipLookupWidth(const HprtTextureDesc &tx, float su, float sv, float width) {
    const int nLevels = (int)tx.n_levels;
    const float invLog2 = 1.442695040888963387004650940071;
    const float level = nLevels - 1 + det_logf(sel_max(width, (float)1e-8)) * invLog2;
    if (level < 0) return HostMipTriangle(tx, 0, su, sv);
    else if (level >= nLevels - 1) return HostMipTexel(tx, nLevels - 1, 0, 0);
    const int iLevel = (int)std::floor(level);
    const float delta = level - iLevel;
    return (1 - delta) * HostMipTriangle(tx, iLevel, su, sv) + delta * HostMipTriangle(tx, iLevel + 1, su, sv);
}
}  // namespace

int hprt_scene_create(const HprtSceneDesc *d, int device, HprtScene **out) try {
    if (!d || !out) return SetError(HPRT_E_INVALID, "hprt_scene_create: null argument");
    if ((d->n_nodes && !d->nodes) || (d->n_prims && !d->prim_order) || (d->n_shapes && !d->shapes) ||
        (d->n_materials && !d->materials) || (d->n_lights && !d->lights))
        return SetError(HPRT_E_INVALID, "hprt_scene_create: null array with non-zero count");
    if (d->n_textures && !d->textures) return SetError(HPRT_E_INVALID, "hprt_scene_create: null texture array with non-zero count");
    if ((d->n_objects && !d->objects) || (d->n_instances && !d->instances) || (d->n_top && !d->top))
        return SetError(HPRT_E_INVALID, "hprt_scene_create: null instancing array with non-zero count");
    // ---- validate the description against what the kernels assume ----
    std::vector<uint64_t> vtxBase(d->n_shapes + 1, 0);
    std::vector<uint32_t> shapePrims(d->n_shapes, 0);
    uint32_t nSpheres = 0;
    for (uint32_t s = 0; s < d->n_shapes; ++s) {
        const HprtShapeDesc &sh = d->shapes[s];
        if (sh.material < 0 || (uint32_t)sh.material >= d->n_materials) return SetError(HPRT_E_INVALID, "shape material index out of range");
        if (sh.area_light >= (int32_t)d->n_lights) return SetError(HPRT_E_INVALID, "shape area light index out of range");
        if (sh.kind == 0) {
            if (sh.n_tris && (!sh.indices || !sh.P)) return SetError(HPRT_E_INVALID, "mesh without indices or positions");
            for (uint64_t i = 0; i < 3ull * sh.n_tris; ++i)
                if (sh.indices[i] < 0 || (uint32_t)sh.indices[i] >= sh.n_verts) return SetError(HPRT_E_INVALID, "mesh vertex index out of range");
            // an emissive mesh owns one light per triangle, consecutive and in face order (core/api.cpp:1609-1636)
            if (sh.area_light >= 0) {
                if ((uint64_t)sh.area_light + sh.n_tris > d->n_lights) return SetError(HPRT_E_INVALID, "emissive mesh: its per-triangle lights exceed the light table");
                for (uint32_t t = 0; t < sh.n_tris; ++t)
                    if (d->lights[sh.area_light + t].type != 2 || d->lights[sh.area_light + t].shape != (int32_t)s)
                        return SetError(HPRT_E_INVALID, "emissive mesh: light area_light + t must be the diffuse area light of its triangle t");
            }
            shapePrims[s] = sh.n_tris; vtxBase[s + 1] = vtxBase[s] + sh.n_verts;
        } else if (sh.kind == 1) {
            shapePrims[s] = 1; vtxBase[s + 1] = vtxBase[s]; ++nSpheres;
        } else return SetError(HPRT_E_INVALID, "unknown shape kind");
    }
    (void)nSpheres;
    if (vtxBase[d->n_shapes] > 0xffffffffull) return SetError(HPRT_E_UNSUPPORTED, "more than 2^32 vertices");
    // ---- aggregates: 0 = the top level (renderOptions->primitives), 1 + k = object definition k ----
    struct AggPrim { int32_t shape; uint32_t local; };      // shape < 0: instance `local`
    struct Agg { const BvhNode *nodes; uint32_t nNodes; const uint32_t *order; uint32_t nPrims; std::vector<AggPrim> prims; };
    std::vector<Agg> aggs(1 + (size_t)d->n_objects);
    std::vector<int32_t> objectOfShape(d->n_shapes, -1);
    aggs[0].nodes = (const BvhNode *)d->nodes; aggs[0].nNodes = d->n_nodes; aggs[0].order = d->prim_order; aggs[0].nPrims = d->n_prims;
    auto addShape = [&](Agg &a, uint32_t s) { for (uint32_t k = 0; k < shapePrims[s]; ++k) a.prims.push_back(AggPrim{(int32_t)s, k}); };
    for (uint32_t k = 0; k < d->n_objects; ++k) {
        const HprtObjectDesc &o = d->objects[k];
        Agg &a = aggs[1 + (size_t)k];
        if ((uint64_t)o.first_shape + o.n_shapes > d->n_shapes) return SetError(HPRT_E_INVALID, "object shape range out of bounds");
        if ((o.n_nodes && !o.nodes) || (o.n_prims && !o.prim_order)) return SetError(HPRT_E_INVALID, "object without its aggregate arrays");
        a.nodes = (const BvhNode *)o.nodes; a.nNodes = o.n_nodes; a.order = o.prim_order; a.nPrims = o.n_prims;
        for (uint32_t s = o.first_shape; s < o.first_shape + o.n_shapes; ++s) {
            if (objectOfShape[s] >= 0) return SetError(HPRT_E_INVALID, "a shape belongs to two objects");
            if (d->shapes[s].area_light >= 0) return SetError(HPRT_E_UNSUPPORTED, "area lights are not supported with object instancing (core/api.cpp:1640)");
            objectOfShape[s] = (int32_t)k;
            addShape(a, s);
        }
    }
    for (uint32_t i = 0; i < d->n_instances; ++i) {
        const int32_t o = d->instances[i].object;
        if (o < 0 || (uint32_t)o >= d->n_objects) return SetError(HPRT_E_INVALID, "instance object index out of range");
        if (aggs[1 + (size_t)o].prims.empty()) return SetError(HPRT_E_INVALID, "instance of an empty object");
    }
    if (d->top) {
        for (uint32_t t = 0; t < d->n_top; ++t) {
            const HprtTopItem &it = d->top[t];
            if (it.kind == 0) {
                if (it.index >= d->n_shapes || objectOfShape[it.index] >= 0) return SetError(HPRT_E_INVALID, "top-level item references a missing or object-owned shape");
                addShape(aggs[0], it.index);
            } else if (it.kind == 1) {
                if (it.index >= d->n_instances) return SetError(HPRT_E_INVALID, "top-level item references a missing instance");
                aggs[0].prims.push_back(AggPrim{-1, it.index});
            } else return SetError(HPRT_E_INVALID, "unknown top-level item kind");
        }
    } else {
        if (d->n_instances) return SetError(HPRT_E_INVALID, "instances need the top-level item list");
        for (uint32_t s = 0; s < d->n_shapes; ++s) if (objectOfShape[s] < 0) addShape(aggs[0], s);
    }
    // primitive records and node pairs of all aggregates share one array each; 32-bit byte offsets address them
    std::vector<uint32_t> primBase(aggs.size() + 1, 0), pairBase(aggs.size() + 1, 0);
    {
        uint64_t np = 0, npair = 0;
        for (size_t a = 0; a < aggs.size(); ++a) {
            const Agg &g = aggs[a];
            if (g.prims.size() != g.nPrims) return SetError(HPRT_E_INVALID, "prim_order length does not match the aggregate's primitive count");
            // BVH sanity: every child/primitive reference must stay inside the arrays
            uint32_t interior = 0;
            for (uint32_t i = 0; i < g.nNodes; ++i) {
                uint32_t axis = g.nodes[i].countAxis & 3u, cnt = g.nodes[i].countAxis >> 2;
                if (axis == 3u) { if (g.nodes[i].offset < 0 || cnt == 0 || (uint64_t)g.nodes[i].offset + cnt > g.nPrims) return SetError(HPRT_E_INVALID, "BVH leaf is empty or references primitives out of range"); }
                else { ++interior; if (g.nodes[i].offset <= (int32_t)i || (uint32_t)g.nodes[i].offset >= g.nNodes || i + 1 >= g.nNodes) return SetError(HPRT_E_INVALID, "BVH interior node has an invalid child"); }
            }
            if ((g.nPrims != 0) != (g.nNodes != 0)) return SetError(HPRT_E_INVALID, "aggregate with primitives but no nodes (or the reverse)");
            for (uint32_t i = 0; i < g.nPrims; ++i) if (g.order[i] >= g.nPrims) return SetError(HPRT_E_INVALID, "prim_order entry out of range");
            primBase[a] = (uint32_t)np; pairBase[a] = (uint32_t)npair;
            np += g.nPrims; npair += g.nNodes ? interior + 1u : 0u;
            if (np * 48ull > 0xffffffffull || npair * 64ull > 0xffffffffull)
                return SetError(HPRT_E_UNSUPPORTED, "more than 89,478,485 primitives (or 67,108,863 interior nodes) over all aggregates");
        }
        primBase[aggs.size()] = (uint32_t)np; pairBase[aggs.size()] = (uint32_t)npair;
    }
    const uint32_t totalPrims = primBase[aggs.size()];
    for (uint32_t l = 0; l < d->n_lights; ++l) {
        const HprtLightDesc &L = d->lights[l];
        if (L.type < 0 || L.type > 3) return SetError(HPRT_E_INVALID, "unknown light type");
        if (L.type == 3) {
            if (L.texture < 0 || (uint32_t)L.texture >= d->n_textures) return SetError(HPRT_E_INVALID, "infinite light: map index out of range");
            const HprtTextureDesc &mt = d->textures[L.texture];
            if (!mt.levels || mt.n_levels == 0 || mt.n_levels > 26 || mt.wrap != 0) return SetError(HPRT_E_INVALID, "infinite light: its map must be a repeat-wrapped pyramid of at most 26 levels");
            if ((uint64_t)mt.levels[0].w * (uint64_t)mt.levels[0].h > (1ull << 26)) return SetError(HPRT_E_UNSUPPORTED, "infinite light: map larger than 2^26 texels");
        }
        if (L.type == 2) {
            if (L.shape < 0 || (uint32_t)L.shape >= d->n_shapes) return SetError(HPRT_E_INVALID, "area light shape out of range");
            const HprtShapeDesc &ls = d->shapes[L.shape];
            const bool own = ls.kind == 1 ? ls.area_light == (int32_t)l
                                          : ls.area_light >= 0 && (int32_t)l >= ls.area_light && (uint32_t)((int32_t)l - ls.area_light) < ls.n_tris;
            if (!own) return SetError(HPRT_E_INVALID, "area light and its shape do not reference each other");
        }
    }
    // CreateLightSampleDistribution (core/lightdistrib.cpp:48-66): a single light always gets the uniform distribution
    const int lightStrategy = d->n_lights <= 1 ? 0 : d->light_strategy;
    if (lightStrategy < 0 || lightStrategy > 2) return SetError(HPRT_E_INVALID, "light_strategy must be 0 (uniform), 1 (power) or 2 (spatial)");
    int dev;
    int rc = CheckDevice(device, &dev);
    if (rc != HPRT_OK) return rc;

    HprtScene *sc = new HprtScene();
    sc->device = dev; sc->nPrims = totalPrims;
    std::unique_ptr<HprtScene> guard(sc);
    // ---- flatten ----
    const uint32_t nVtx = (uint32_t)vtxBase[d->n_shapes];
    std::vector<float> vUV(2 * (size_t)nVtx, 0.f), vS(3 * (size_t)nVtx, 0.f);
    std::vector<DevShape> shapes(d->n_shapes);
    std::vector<DevSphere> spheres; std::vector<int> sphereOfShape(d->n_shapes, -1);
    for (uint32_t s = 0; s < d->n_shapes; ++s) {
        const HprtShapeDesc &sh = d->shapes[s];
        DevShape &o = shapes[s];
        o.material = sh.material; o.areaLight = sh.area_light; o.sphere = -1;
        o.flags = ((sh.reverse_orientation != 0) ^ (sh.transform_swaps_handedness != 0)) ? SHAPE_FLIP : 0u;
        if (sh.reverse_orientation) o.flags |= SHAPE_REVERSE;
        if (sh.kind == 0) {
            size_t b = vtxBase[s];
            if (sh.N) o.flags |= SHAPE_HAS_N;
            if (sh.UV) { o.flags |= SHAPE_HAS_UV; memcpy(&vUV[2 * b], sh.UV, 8 * (size_t)sh.n_verts); }
            if (sh.S) { o.flags |= SHAPE_HAS_S; memcpy(&vS[3 * b], sh.S, 12 * (size_t)sh.n_verts); }
        } else {
            DevSphere sp;
            memcpy(sp.o2w.m, sh.object_to_world, 64); memcpy(sp.w2o.m, sh.world_to_object, 64);
            sp.radius = sh.radius; sp.zMin = sh.z_min; sp.zMax = sh.z_max; sp.thetaMin = sh.theta_min; sp.thetaMax = sh.theta_max; sp.phiMax = sh.phi_max;
            o.sphere = sphereOfShape[s] = (int)spheres.size();
            spheres.push_back(sp);
        }
    }
    std::vector<float4> tris(3 * (size_t)totalPrims);
    std::vector<uint32_t> primVtx(3 * (size_t)totalPrims, 0u);
    std::vector<float4> primN(3 * (size_t)totalPrims, make_float4(0.f, 0.f, 0.f, 0.f));
    std::vector<int32_t> lightPrim(d->n_lights, -1);      // triangle lights: ordered index of their triangle
    for (size_t ai = 0; ai < aggs.size(); ++ai) {
        const Agg &g = aggs[ai];
        for (uint32_t oi = 0; oi < g.nPrims; ++oi) {
            const size_t i = (size_t)primBase[ai] + oi;
            const AggPrim e = g.prims[g.order[oi]];
            float4 r0, r1, r2;
            if (e.shape < 0) {     // TransformedPrimitive
                r0 = make_float4(0, 0, 0, u2f(TAG_INSTANCE)); r1 = make_float4(0, 0, 0, u2f(0u)); r2 = make_float4(0, 0, 0, u2f(e.local));
            } else {
                const uint32_t s = (uint32_t)e.shape;
                const HprtShapeDesc &sh = d->shapes[s];
                if (sh.kind == 0) {
                    const int32_t *v = &sh.indices[3 * (size_t)e.local];
                    const float *a = &sh.P[3 * (size_t)v[0]], *b = &sh.P[3 * (size_t)v[1]], *c = &sh.P[3 * (size_t)v[2]];
                    bool bogus = TriangleIsBogus(vec3(a[0], a[1], a[2]), vec3(b[0], b[1], b[2]), vec3(c[0], c[1], c[2]),
                                                 sh.UV ? &sh.UV[2 * (size_t)v[0]] : nullptr, sh.UV ? &sh.UV[2 * (size_t)v[1]] : nullptr,
                                                 sh.UV ? &sh.UV[2 * (size_t)v[2]] : nullptr);
                    const HprtMaterialDesc &md = d->materials[sh.material];
                    // a triangle of an emissive mesh: aux = 1 + its light, and TAG_GENERIC — the generic shading variant is the one that looks for Le
                    const int32_t triLight = sh.area_light >= 0 ? sh.area_light + (int32_t)e.local : -1;
                    if (triLight >= 0) lightPrim[triLight] = (int32_t)i;
                    const bool textured = md.kd_texture >= 0 || md.ks_texture >= 0 || (md.type == 6 && md.opacity_texture >= 0);
                    // the shading bin: plain matte / plastic / substrate triangles have kernels of their own; emitters, textured and every other
                    // material (OrenNayar, mirror, metal, glass, uber) are shaded by the generic variant
                    const uint32_t bin = textured ? BIN_TEXTURED : triLight >= 0 ? BIN_GENERIC : md.type == 1 ? BIN_PLASTIC : md.type == 3 ? BIN_SUBSTRATE :
                                         (md.type == 0 && clampf(md.sigma, 0.f, 90.f) == 0.f) ? BIN_MATTE : BIN_GENERIC;
                    uint32_t tag = (bogus ? TAG_BOGUS : 0u) | (bin << TAG_BIN_SHIFT);
                    if (bin == BIN_SUBSTRATE) sc->hasSubstrateBin = true;
                    r0 = make_float4(a[0], a[1], a[2], u2f(tag)); r1 = make_float4(b[0], b[1], b[2], u2f(s)); r2 = make_float4(c[0], c[1], c[2], u2f((uint32_t)(triLight + 1)));
                    for (int k = 0; k < 3; ++k) {
                        primVtx[3 * i + k] = (uint32_t)(vtxBase[s] + (uint32_t)v[k]);
                        if (sh.N) { const float *nn = &sh.N[3 * (size_t)v[k]]; primN[3 * i + k] = make_float4(nn[0], nn[1], nn[2], 0.f); }
                    }
                } else {
                    const HprtMaterialDesc &md = d->materials[sh.material];
                    const bool textured = md.kd_texture >= 0 || md.ks_texture >= 0 || (md.type == 6 && md.opacity_texture >= 0);
                    r0 = make_float4(0, 0, 0, u2f(TAG_SPHERE | ((textured ? BIN_TEXTURED : BIN_GENERIC) << TAG_BIN_SHIFT))); r1 = make_float4(0, 0, 0, u2f(s)); r2 = make_float4(0, 0, 0, u2f((uint32_t)sphereOfShape[s]));
                }
            }
            tris[3 * i] = r0; tris[3 * i + 1] = r1; tris[3 * i + 2] = r2;
        }
    }
    std::vector<DevMaterial> mats(d->n_materials);
    for (uint32_t m = 0; m < d->n_materials; ++m) {
        const HprtMaterialDesc &in = d->materials[m];
        if (in.type < 0 || in.type > 6) return SetError(HPRT_E_UNSUPPORTED, "material type outside the hot-path scope (matte, plastic, mirror, substrate, metal, glass, uber)");
        if (in.kd_texture >= (int32_t)d->n_textures || in.ks_texture >= (int32_t)d->n_textures || (in.type == 6 && in.opacity_texture >= (int32_t)d->n_textures))
            return SetError(HPRT_E_INVALID, "material texture index out of range");
        DevMaterial &o = mats[m];
        o.KdTex = in.kd_texture >= 0 ? in.kd_texture : -1; o.KsTex = in.ks_texture >= 0 ? in.ks_texture : -1;
        o.opTex = in.type == 6 && in.opacity_texture >= 0 ? in.opacity_texture : -1;
        o.type = in.type; memcpy(o.Kd, in.Kd, 12); memcpy(o.Ks, in.Ks, 12);
        o.alpha = in.remap_roughness ? RoughnessToAlpha(in.roughness) : in.roughness;
        o.alphaY = o.alpha;
        if (in.type == 3 || in.type == 4 || in.type == 6) o.alphaY = in.remap_roughness ? RoughnessToAlpha(in.sigma) : in.sigma;      // substrate / metal / uber: sigma carries vroughness
        memcpy(o.Kr, in.Kr, 12); memcpy(o.Kt, in.Kt, 12); memcpy(o.opacity, in.opacity, 12); o.eta = in.eta;
        o.roughGlass = 0;
        if (in.type == 5) {
            o.alpha = in.roughness;      // glass: the index of refraction, as given
            // rough dielectric (materials/glass.cpp:61-72): isSpecular is decided on the values as given, the remap applies to both after it
            if (in.sigma != 0.f || in.Kr[0] != 0.f) {
                o.roughGlass = 1;
                o.Kr[0] = in.remap_roughness ? RoughnessToAlpha(in.sigma) : in.sigma;
                o.Kr[1] = in.remap_roughness ? RoughnessToAlpha(in.Kr[0]) : in.Kr[0];
            }
        }
        // MatteMaterial: sig = Clamp(sigma, 0, 90); sig != 0 -> OrenNayar(r, sig) (materials/matte.cpp:55-61, core/reflection.h:414-420)
        const float sig = clampf(in.sigma, 0.f, 90.f);
        o.oren = in.type == 0 && sig != 0.f ? 1 : 0; o.orenA = 1.f; o.orenB = 0.f;
        if (o.oren) {
            const float sigma = (HPRT_PI / 180) * sig;
            const float sigma2 = sigma * sigma;
            o.orenA = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
            o.orenB = 0.45f * sigma2 / (sigma2 + 0.09f);
        }
    }
    // image textures: every pyramid level of every texture in one texel array (3 floats per texel)
    std::vector<DevTexture> textures(d->n_textures);
    std::vector<DevMipLevel> mipLevels;
    std::vector<float> texels, weightLut;
    for (uint32_t t = 0; t < d->n_textures; ++t) {
        const HprtTextureDesc &in = d->textures[t];
        if (!in.levels || in.n_levels == 0 || in.n_levels > 32 || !in.weight_lut) return SetError(HPRT_E_INVALID, "texture without levels or weight table");
        if (in.wrap < 0 || in.wrap > 2) return SetError(HPRT_E_INVALID, "texture wrap mode out of range");
        DevTexture &o = textures[t];
        o.firstLevel = (uint32_t)mipLevels.size(); o.nLevels = in.n_levels; o.trilinear = in.trilinear ? 1 : 0; o.wrap = in.wrap;
        o.maxAniso = in.max_anisotropy; o.su = in.su; o.sv = in.sv; o.du = in.du; o.dv = in.dv;
        for (uint32_t l = 0; l < in.n_levels; ++l) {
            const HprtTextureLevel &lv = in.levels[l];
            if (lv.w <= 0 || lv.h <= 0 || !lv.rgb) return SetError(HPRT_E_INVALID, "empty texture level");
            const size_t n = 3 * (size_t)lv.w * (size_t)lv.h;
            if (texels.size() + n > 0xffffffffull) return SetError(HPRT_E_UNSUPPORTED, "more than 2^32 texture floats");
            mipLevels.push_back(DevMipLevel{(uint32_t)texels.size(), lv.w, lv.h});
            texels.insert(texels.end(), lv.rgb, lv.rgb + n);
        }
        // MIPMap::weightLut is a static table (core/mipmap.h:107, 153-161): identical for every texture
        if (t == 0) weightLut.assign(in.weight_lut, in.weight_lut + 128);
        else if (memcmp(weightLut.data(), in.weight_lut, 128 * sizeof(float)) != 0) return SetError(HPRT_E_INVALID, "textures disagree on the EWA weight table");
    }
    std::vector<DevLight> lights(d->n_lights);
    for (uint32_t l = 0; l < d->n_lights; ++l) {
        const HprtLightDesc &in = d->lights[l];
        DevLight &o = lights[l];
        o.type = in.type; memcpy(o.pos, in.pos, 12); memcpy(o.I, in.I, 12); o.shape = in.shape; o.twoSided = in.two_sided;
        const bool onMesh = in.type == 2 && d->shapes[in.shape].kind == 0;
        if (onMesh && lightPrim[l] < 0) return SetError(HPRT_E_UNSUPPORTED, "emissive triangle outside the top-level aggregate (area lights are not supported with object instancing, core/api.cpp:1640)");
        if (onMesh) o.type = 3;
        o.prim = onMesh ? lightPrim[l] : -1;
        o.sphere = in.type == 2 && !onMesh ? sphereOfShape[in.shape] : -1;
        o.shapeFlags = in.type == 2 ? shapes[in.shape].flags : 0u;
    }
    // infinite lights: the scalar image of lights/infinite.cpp:66-85 (2w x 2h: luminance of the filtered map times sin theta) and
    // its Distribution2D, built with the float operations of Distribution1D's constructor (hprt_math.h dist1d_build)
    std::vector<DevEnvLight> envLights;
    std::vector<float> envData;
    for (uint32_t l = 0; l < d->n_lights; ++l) {
        const HprtLightDesc &in = d->lights[l];
        if (in.type != 3) continue;
        const HprtTextureDesc &tx = d->textures[in.texture];
        DevEnvLight e; memset(&e, 0, sizeof(e));
        memcpy(&e.l2w, in.light_to_world, 64); memcpy(&e.w2l, in.world_to_light, 64);
        e.tex = in.texture;
        const int width = 2 * tx.levels[0].w, height = 2 * tx.levels[0].h;
        e.nu = width; e.nv = height; e.off = (uint32_t)envData.size();
        const size_t nFloats = (size_t)height * width + (size_t)height * (width + 1) + (size_t)height + (size_t)height + 1;
        if (envData.size() + nFloats > 0x7fffffffull) return SetError(HPRT_E_UNSUPPORTED, "infinite light maps too large");
        envData.resize(envData.size() + nFloats);
        float *condFunc = envData.data() + e.off, *condCdf = condFunc + (size_t)height * width, *condInt = condCdf + (size_t)height * (width + 1),
              *margCdf = condInt + height;
        const float fwidth = 0.5f / std::min(width, height);
        for (int v = 0; v < height; ++v) {
            const float vp = (v + .5f) / (float)height;
            const float sinTheta = det_sinf(HPRT_PI * (v + .5f) / height);
            for (int u = 0; u < width; ++u) {
                const float up = (u + .5f) / (float)width;
                float y = luminance(HostMipLookupWidth(tx, up, vp, fwidth));
                y *= sinTheta;
                condFunc[(size_t)v * width + u] = y;
            }
            dist1d_build(condFunc + (size_t)v * width, width, condCdf + (size_t)v * (width + 1), &condInt[v]);
        }
        dist1d_build(condInt, height, margCdf, &e.margFuncInt);
        DevLight &o = lights[l];
        o.type = 4; o.shape = (int32_t)envLights.size();
        envLights.push_back(e);
    }
    // Scene::worldBound + Bounds3::BoundingSphere (core/scene.h:56-66, core/geometry.h:980-983)
    float worldRadius = 0.f;
    vec3 wbLo, wbHi;
    if (d->n_nodes) {
        const BvhNode &root = ((const BvhNode *)d->nodes)[0];
        wbLo = vec3(root.bmin[0], root.bmin[1], root.bmin[2]); wbHi = vec3(root.bmax[0], root.bmax[1], root.bmax[2]);
        vec3 c = div_by(wbLo + wbHi, 2.f);
        bool inside = c.x >= wbLo.x && c.x <= wbHi.x && c.y >= wbLo.y && c.y <= wbHi.y && c.z >= wbLo.z && c.z <= wbHi.z;
        worldRadius = inside ? dist(c, wbHi) : 0.f;
    }
    // UniformLightDistribution (core/lightdistrib.cpp:68-75) or PowerLightDistribution (:77-82, ComputeLightPowerDistribution,
    // core/integrator.cpp:219-227: Light::Power().y()) as a Distribution1D (core/sampling.h:57-70)
    std::vector<float> func(std::max<uint32_t>(1, d->n_lights), 1.f), cdf(d->n_lights + 1, 0.f);
    if (lightStrategy == 1)
        for (uint32_t l = 0; l < d->n_lights; ++l) {
            const HprtLightDesc &in = d->lights[l];
            const rgb I(in.I[0], in.I[1], in.I[2]);
            rgb power;
            if (in.type == 0) power = I * (4 * HPRT_PI);                                      // lights/point.cpp:55
            else if (in.type == 1) power = I * HPRT_PI * worldRadius * worldRadius;          // lights/distant.cpp:61-63
            else if (in.type == 3) power = HPRT_PI * worldRadius * worldRadius * HostMipLookupWidth(d->textures[in.texture], .5f, .5f, .5f);   // lights/infinite.cpp:87-91
            else {                                                                            // lights/diffuse.cpp:64-66, area = shape->Area()
                const HprtShapeDesc &ls = d->shapes[in.shape];
                float area;
                if (ls.kind == 1) area = ls.phi_max * ls.radius * (ls.z_max - ls.z_min);      // Sphere::Area, shapes/sphere.cpp:215
                else {                                                                        // Triangle::Area, shapes/triangle.cpp:576-582
                    const int32_t *v = &ls.indices[3 * (size_t)((int32_t)l - ls.area_light)];
                    const vec3 p0(ls.P[3 * v[0]], ls.P[3 * v[0] + 1], ls.P[3 * v[0] + 2]), p1(ls.P[3 * v[1]], ls.P[3 * v[1] + 1], ls.P[3 * v[1] + 2]),
                               p2(ls.P[3 * v[2]], ls.P[3 * v[2] + 1], ls.P[3 * v[2] + 2]);
                    area = (float)(0.5 * (double)length(cross(p1 - p0, p2 - p0)));
                }
                power = I * (float)(in.two_sided ? 2 : 1) * area * HPRT_PI;
            }
            func[l] = luminance(power);
        }
    float funcInt = 0.f;
    if (d->n_lights) dist1d_build(func.data(), (int)d->n_lights, cdf.data(), &funcInt);
    // Halton tables + 64-bit division magics
    const std::vector<uint16_t> &perms = HaltonPermutations();
    // ---- child-pair layout of the BVHs (device/dev_scene.h): one run of pairs per aggregate ----
    std::vector<DevPair> pairs((size_t)pairBase[aggs.size()]);
    std::vector<DevWide> wide;
    std::vector<float4> leafBox;
    std::vector<int32_t> wideBase(aggs.size(), -1);      // first wide record of every aggregate that has a tree
    bool wideOk = true; int wideNeedTop = 0, wideNeedObject = 0;
    // The ordered walk keeps at most one pending sibling per level, plus the sentinel of an instance: the kernel's
    // stack has HPRT_STACK_TOTAL = 64 entries (LDS + HBM part), as the reference's nodesToVisit[64]
    // (accelerators/bvh.cpp:365).  Deeper trees are refused here rather than walked wrongly.
    int topDepth = 0, objectDepth = 0;
    for (size_t ai = 0; ai < aggs.size(); ++ai) {
        const Agg &g = aggs[ai];
        if (g.nNodes == 0) continue;
        const BvhNode *nd = g.nodes;
        {   // depth of this aggregate's tree (depth-first layout: first child at i + 1, second child at offset)
            std::vector<std::pair<uint32_t, int>> todo; todo.push_back({0u, 1});
            int depth = 0;
            while (!todo.empty()) {
                auto [i, dpt] = todo.back(); todo.pop_back();
                if (i >= g.nNodes) return SetError(HPRT_E_INVALID, "BVH node index out of range");
                depth = std::max(depth, dpt);
                if ((nd[i].countAxis & 3u) != 3u) {
                    if ((uint32_t)nd[i].offset <= i || (uint32_t)nd[i].offset >= g.nNodes) return SetError(HPRT_E_INVALID, "BVH second-child offset out of range");
                    todo.push_back({(uint32_t)nd[i].offset, dpt + 1}); todo.push_back({i + 1u, dpt + 1});
                }
            }
            if (ai == 0) topDepth = depth; else objectDepth = std::max(objectDepth, depth);
        }
        std::vector<int32_t> ref(g.nNodes);
        int32_t nextPair = (int32_t)pairBase[ai] + 1;
        for (uint32_t i = 0; i < g.nNodes; ++i) {
            if ((nd[i].countAxis & 3u) == 3u) {
                ref[i] = ~(int32_t)(primBase[ai] + (uint32_t)nd[i].offset);
                const size_t last = (size_t)primBase[ai] + (uint32_t)nd[i].offset + (nd[i].countAxis >> 2) - 1u;
                tris[3 * last].w = u2f(f2u(tris[3 * last].w) | TAG_LAST);
            } else ref[i] = nextPair++;
        }
        auto fill = [&](DevPair &p, uint32_t c0, uint32_t c1, uint32_t meta) {
            const BvhNode &a = nd[c0], &b = nd[c1];
            p.x[0] = a.bmin[0]; p.x[1] = b.bmin[0]; p.x[2] = a.bmax[0]; p.x[3] = b.bmax[0];
            p.y[0] = a.bmin[1]; p.y[1] = b.bmin[1]; p.y[2] = a.bmax[1]; p.y[3] = b.bmax[1];
            p.z[0] = a.bmin[2]; p.z[1] = b.bmin[2]; p.z[2] = a.bmax[2]; p.z[3] = b.bmax[2];
            p.ref0 = ref[c0]; p.ref1 = ref[c1]; p.meta = meta; p.pad = 0u;
        };
        fill(pairs[pairBase[ai]], 0u, 0u, PAIR_SINGLE);      // synthetic parent of the root: carries the root's bounds test
        for (uint32_t i = 0; i < g.nNodes; ++i) {
            if ((nd[i].countAxis & 3u) == 3u) continue;
            const uint32_t c[2] = {i + 1u, (uint32_t)nd[i].offset};
            fill(pairs[(size_t)ref[i]], c[0], c[1], nd[i].countAxis & 3u);
        }
        // The leaf-exact walk of plain renders (wide_bvh.h): four-wide records over the same leaves, for scenes without object instances.
        // A leaf that holds exactly one triangle needs no stored box: Triangle::WorldBound is the min / max of its vertices
        // (shapes/triangle.cpp:180-186) — provided that is, bit for bit, what the node holds (a zero of either sign among the
        // coordinates would make the minimum's sign a matter of operand order: such leaves read their box like the others).
        if (wideOk && totalPrims < (1u << 28)) {
            std::vector<int32_t> leafRefW(g.nNodes, WIDE_NONE);
            if (leafBox.empty()) leafBox.assign(2 * (size_t)totalPrims, make_float4(0.f, 0.f, 0.f, 0.f));
            for (uint32_t i = 0; i < g.nNodes; ++i) {
                if ((nd[i].countAxis & 3u) != 3u) continue;
                const uint32_t firstPrim = primBase[ai] + (uint32_t)nd[i].offset, count = nd[i].countAxis >> 2;
                bool single = count == 1u && (f2u(tris[3 * (size_t)firstPrim].w) & TAG_KIND_MASK) == 0u;
                if (single) {
                    const float4 *v = &tris[3 * (size_t)firstPrim];
                    const float c[3][3] = {{v[0].x, v[1].x, v[2].x}, {v[0].y, v[1].y, v[2].y}, {v[0].z, v[1].z, v[2].z}};
                    for (int a = 0; a < 3 && single; ++a) {
                        bool posZero = false, negZero = false;
                        for (int k = 0; k < 3; ++k) { if (c[a][k] != c[a][k]) single = false; if (c[a][k] == 0.f) { if (f2u(c[a][k]) >> 31) negZero = true; else posZero = true; } }
                        const float mn = std::min(std::min(c[a][0], c[a][1]), c[a][2]), mx = std::max(std::max(c[a][0], c[a][1]), c[a][2]);
                        if ((posZero && negZero) || f2u(mn) != f2u(nd[i].bmin[a]) || f2u(mx) != f2u(nd[i].bmax[a])) single = false;
                    }
                }
                uint32_t r = ~firstPrim;
                if (!single) r &= ~WIDE_LEAF_BOXED;
                leafRefW[i] = (int32_t)r;
                for (uint32_t k = 0; k < count; ++k) {
                    leafBox[2 * (size_t)(firstPrim + k)] = make_float4(nd[i].bmin[0], nd[i].bmin[1], nd[i].bmin[2], nd[i].bmax[0]);
                    leafBox[2 * (size_t)(firstPrim + k) + 1] = make_float4(nd[i].bmax[1], nd[i].bmax[2], 0.f, 0.f);
                }
            }
            int need = 0;
            wideBase[ai] = (int32_t)wide.size();
            if (!BuildWide(nd, g.nNodes, leafRefW.data(), &wide, &need)) wideOk = false;
            if (ai == 0) wideNeedTop = need; else wideNeedObject = std::max(wideNeedObject, need);
        } else wideOk = false;
    }
    // (the wide walk's stack: LDS entries + the scene's deep-stack area; a scene that could need more keeps the binary walk)
    if (!wideOk || wideNeedTop + (d->n_instances ? 1 + wideNeedObject : 0) > HPRT_WIDE_STACK_MAX || wideBase[0] != 0) { wide.clear(); leafBox.clear(); }
    if (topDepth + (d->n_instances ? 1 + objectDepth : 0) > HPRT_STACK_TOTAL)
        return SetError(HPRT_E_UNSUPPORTED, "BVH deeper than the 64-entry traversal stack (accelerators/bvh.cpp:365 reserves the same)");
    std::vector<DevInstance> instances(d->n_instances);
    for (uint32_t i = 0; i < d->n_instances; ++i) {
        const HprtInstanceDesc &in = d->instances[i];
        DevInstance &o = instances[i];
        memcpy(o.i2w.m, in.instance_to_world, 64); memcpy(o.w2i.m, in.world_to_instance, 64);
        const size_t ai = 1 + (size_t)in.object;
        // more than one primitive: the object's aggregate; one: that primitive itself, without a bounds test (core/api.cpp:1798-1806)
        o.root = aggs[ai].nPrims > 1 ? (int32_t)pairBase[ai] : ~(int32_t)primBase[ai];
        bool ident = true;                                   // Transform::IsIdentity, core/transform.h:148-155
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) if (o.i2w.m[r][c] != (r == c ? 1.f : 0.f)) ident = false;
        o.identity = ident ? 1u : 0u; o.pad[0] = o.pad[1] = 0u;
        // the entry into the wide records (k_walk4): the object's own tree, or its one primitive as a leaf "already reached" (no bounds test)
        if (!wide.empty()) o.pad[0] = aggs[ai].nPrims > 1 ? (uint32_t)wideBase[ai] : ((~(uint32_t)primBase[ai]) & ~WIDE_LEAF_FIRST);
    }
    // Instance primitives of the top level: the transform moves next to the primitive (dev_scene.h, TAG_INST_INLINE / topEntry)
    std::vector<float4> topEntry;
    if (d->n_instances) {
        topEntry.assign(aggs[0].nPrims, make_float4(0.f, 0.f, 0.f, 0.f));
        for (uint32_t oi = 0; oi < aggs[0].nPrims; ++oi) {
            const size_t i = (size_t)primBase[0] + oi;
            const uint32_t tag = f2u(tris[3 * i].w);
            if ((tag & TAG_KIND_MASK) != TAG_INSTANCE) continue;
            const DevInstance &in = instances[f2u(tris[3 * i + 2].w)];
            const float (*m)[4] = in.w2i.m;
            if (!(m[3][0] == 0.f && m[3][1] == 0.f && m[3][2] == 0.f && m[3][3] == 1.f)) continue;      // projective: the kernel reads DevInstance
            tris[3 * i] = make_float4(m[0][0], m[0][1], m[0][2], u2f(tag | TAG_INST_INLINE));
            tris[3 * i + 1] = make_float4(m[1][0], m[1][1], m[1][2], tris[3 * i + 1].w);
            tris[3 * i + 2] = make_float4(m[2][0], m[2][1], m[2][2], tris[3 * i + 2].w);
            topEntry[oi] = make_float4(m[0][3], m[1][3], m[2][3], u2f((uint32_t)in.root));
        }
    }
    std::vector<float4> topEntryWide;
    if (d->n_instances && !wide.empty()) {
        topEntryWide = topEntry;
        for (uint32_t oi = 0; oi < aggs[0].nPrims; ++oi) {
            const size_t i = (size_t)primBase[0] + oi;
            const uint32_t tag = f2u(tris[3 * i].w);
            if ((tag & TAG_KIND_MASK) == TAG_INSTANCE && (tag & TAG_INST_INLINE)) topEntryWide[oi].w = u2f(instances[f2u(tris[3 * i + 2].w)].pad[0]);
        }
    }
    std::vector<int32_t> primes(PrimeTable().begin(), PrimeTable().end()), primeSums(PrimeSumTable().begin(), PrimeSumTable().end());
    std::vector<uint64_t> magic(primes.size());
    for (size_t i = 0; i < primes.size(); ++i) magic[i] = 0xffffffffffffffffull / (uint64_t)primes[i] + 1ull;
    // ---- upload ----
    HIP_TRY(upload(sc->nodes, pairs)); HIP_TRY(upload(sc->tris, tris)); HIP_TRY(upload(sc->primVtx, primVtx));
    HIP_TRY(upload(sc->primN, primN)); HIP_TRY(upload(sc->vUV, vUV)); HIP_TRY(upload(sc->vS, vS));
    HIP_TRY(upload(sc->shapes, shapes)); HIP_TRY(upload(sc->materials, mats)); HIP_TRY(upload(sc->lights, lights));
    HIP_TRY(upload(sc->spheres, spheres)); HIP_TRY(upload(sc->instances, instances)); HIP_TRY(upload(sc->topEntry, topEntry)); HIP_TRY(upload(sc->topEntryWide, topEntryWide)); HIP_TRY(upload(sc->lightFunc, func)); HIP_TRY(upload(sc->lightCdf, cdf));
    HIP_TRY(upload(sc->perms, perms)); HIP_TRY(upload(sc->primes, primes)); HIP_TRY(upload(sc->primeSums, primeSums));
    HIP_TRY(upload(sc->primeMagic, magic));
    HIP_TRY(upload(sc->textures, textures)); HIP_TRY(upload(sc->mipLevels, mipLevels)); HIP_TRY(upload(sc->texels, texels)); HIP_TRY(upload(sc->weightLut, weightLut));
    HIP_TRY(upload(sc->wide, wide)); HIP_TRY(upload(sc->leafBox, leafBox));
    HIP_TRY(sc->counters.alloc(sizeof(DevCounters)));
    HIP_TRY(hipMemset(sc->counters.p, 0, sizeof(DevCounters)));
    HIP_TRY(sc->deepStack.alloc((size_t)HPRT_SPILL_STACK * HPRT_DEEP_THREADS * sizeof(uint2)));
    HIP_TRY(sc->workCounter.alloc(256));
    DevScene &dv = sc->dev;
    dv.pairs = sc->nodes.as<DevPair>(); dv.nPairs = (uint32_t)pairs.size();
    dv.wide = wide.empty() ? nullptr : sc->wide.as<DevWide>(); dv.nWide = (uint32_t)wide.size(); dv.leafBox = sc->leafBox.as<float4>();
    dv.tris = sc->tris.as<float4>(); dv.nPrims = totalPrims;
    dv.primVtx = sc->primVtx.as<uint32_t>();
    dv.primN = sc->primN.as<float4>(); dv.vUV = sc->vUV.as<float>(); dv.vS = sc->vS.as<float>();
    dv.shapes = sc->shapes.as<DevShape>(); dv.nShapes = d->n_shapes;
    dv.materials = sc->materials.as<DevMaterial>();
    dv.lights = sc->lights.as<DevLight>(); dv.nLights = d->n_lights;
    dv.spheres = sc->spheres.as<DevSphere>(); dv.nSpheres = (uint32_t)spheres.size();
    HIP_TRY(upload(sc->envLights, envLights)); HIP_TRY(upload(sc->envData, envData));
    dv.envLights = sc->envLights.as<DevEnvLight>(); dv.envData = sc->envData.as<float>(); dv.nEnvLights = (uint32_t)envLights.size();
    dv.textures = d->n_textures ? sc->textures.as<DevTexture>() : nullptr; dv.mipLevels = sc->mipLevels.as<DevMipLevel>();
    dv.texels = sc->texels.as<float>(); dv.weightLut = sc->weightLut.as<float>();
    dv.instances = sc->instances.as<DevInstance>(); dv.nInstances = d->n_instances;
    dv.topEntry = topEntry.empty() ? nullptr : sc->topEntry.as<float4>(); dv.nTopPrims = (uint32_t)topEntry.size();
    dv.topEntryWide = topEntryWide.empty() ? nullptr : sc->topEntryWide.as<float4>();
    dv.lightFunc = sc->lightFunc.as<float>(); dv.lightCdf = sc->lightCdf.as<float>(); dv.lightFuncInt = funcInt;
    dv.deepStack = sc->deepStack.as<uint2>();
    dv.perms = sc->perms.as<uint16_t>(); dv.primes = sc->primes.as<int32_t>(); dv.primeSums = sc->primeSums.as<int32_t>();
    dv.primeMagic = sc->primeMagic.as<uint64_t>();
    dv.worldRadius = worldRadius;
    dv.spatial = 0; dv.voxN[0] = dv.voxN[1] = dv.voxN[2] = 1; dv.voxFunc = dv.voxCdf = dv.voxFuncInt = nullptr;
    dv.voxSlot = nullptr; dv.voxRequest = nullptr; dv.voxRequestCount = nullptr;
    for (int a = 0; a < 3; ++a) { dv.wbMin[a] = wbLo.get(a); dv.wbMax[a] = wbHi.get(a); }
    if (lightStrategy == 2) {
        // SpatialLightDistribution (core/lightdistrib.cpp:95-120, maxVoxels = 64): the voxel grid over the world bound.  The reference
        // fills a voxel's distribution when a vertex first falls into it; a voxel's distribution being a pure function of the voxel,
        // here every voxel is computed now, on the device (k_voxel_contrib / k_voxel_dist).
        const vec3 diag = wbHi - wbLo;
        const int me = (diag.x > diag.y && diag.x > diag.z) ? 0 : (diag.y > diag.z ? 1 : 2);      // Bounds3::MaximumExtent
        const float bmax = diag.get(me);
        uint64_t nVox = 1;
        for (int a = 0; a < 3; ++a) { dv.voxN[a] = std::max(1, int(std::round(diag.get(a) / bmax * 64))); nVox *= (uint64_t)dv.voxN[a]; }
        // The table of every voxel (2 * nLights + 2 floats each) is computed now when it is small: a lookup is then a plain read.  With
        // many lights (every triangle of an emissive mesh is one, core/api.cpp:1609-1636) it is not — 64^3 voxels x 500 lights is
        // already 1 GiB — and the reference never builds it either: it fills a voxel when a vertex first falls into it
        // (core/lightdistrib.cpp:149-229).  Same here then, per batch: a bounded pool of rows, a voxel -> row map, and the
        // bounce loop computes the rows its vertices asked for (RunBatch).  HPRT_VOXEL_DENSE_MAX_MB moves the switch (tests: 0).
        const uint64_t denseMaxFloats = [] { const char *e = getenv("HPRT_VOXEL_DENSE_MAX_MB"); return e ? (uint64_t)atoll(e) * (1ull << 18) : (1ull << 28); }();      // (read per scene: tests switch it)
        const uint64_t rowFloats = 2ull * d->n_lights + 2ull;
        const bool dense = nVox * rowFloats <= denseMaxFloats;
        uint64_t rows = nVox;
        if (!dense) {
            // pool: up to 4 GiB of rows (HPRT_VOXEL_POOL_MB), never more than there are voxels, at least one
            const uint64_t poolFloats = [] { const char *e = getenv("HPRT_VOXEL_POOL_MB"); return (e ? (uint64_t)atoll(e) : 4096ull) * (1ull << 18); }();
            rows = std::max<uint64_t>(1, std::min<uint64_t>(nVox, poolFloats / rowFloats));
        }
        HIP_TRY(sc->voxFunc.alloc(rows * d->n_lights * sizeof(float)));
        HIP_TRY(sc->voxCdf.alloc(rows * (d->n_lights + 1ull) * sizeof(float)));
        HIP_TRY(sc->voxFuncInt.alloc(rows * sizeof(float)));
        std::vector<float> ri(5 * 128);
        VoxelSamplePoints(ri.data());
        HIP_TRY(upload(sc->voxRi, ri));
        dv.spatial = 1;
        dv.voxFunc = sc->voxFunc.as<float>(); dv.voxCdf = sc->voxCdf.as<float>(); dv.voxFuncInt = sc->voxFuncInt.as<float>();
        sc->nVoxels = (uint32_t)nVox; sc->voxRows = (uint32_t)rows; sc->voxRowsUsed = 0;
        if (dense) LaunchVoxelDistributions(nullptr, dv, sc->voxRi.as<float>(), (uint32_t)nVox, sc->voxFunc.as<float>(), sc->voxCdf.as<float>(), sc->voxFuncInt.as<float>());
        else {
            HIP_TRY(sc->voxSlot.alloc(nVox * sizeof(int32_t)));
            HIP_TRY(hipMemset(sc->voxSlot.p, 0xff, nVox * sizeof(int32_t)));      // VOX_EMPTY
            HIP_TRY(sc->voxRequest.alloc(nVox * sizeof(uint32_t)));
            HIP_TRY(sc->voxRequestCount.alloc(256));
            HIP_TRY(hipMemset(sc->voxRequestCount.p, 0, 256));
            dv.voxSlot = sc->voxSlot.as<int32_t>(); dv.voxRequest = sc->voxRequest.as<uint32_t>(); dv.voxRequestCount = sc->voxRequestCount.as<uint32_t>();
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
    }
    HIP_TRY(hipHostMalloc((void **)&sc->hostCounts, (4096 + 256) * sizeof(uint32_t)));
    *out = guard.release();
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }

// Diagnostics hook (not part of include/hprt.h): the 128 sample points of a voxel, RadicalInverse(0..4, i) as [5][128]
// Diagnostics hook (not part of include/hprt.h): 1 when plain renders of this scene take the leaf-exact wide walk (k_walk4), 0 for the binary walk
__attribute__((visibility("default"))) int hprt_debug_scene_walk(HprtScene *s) { return s && hprt::WideWalkInUse(s->dev) ? 1 : 0; }
__attribute__((visibility("default"))) int hprt_debug_poison_workspace(HprtScene *s, int byte) { if (!s) return HPRT_E_INVALID; s->poisonByte = byte < 0 ? -1 : (byte & 255); return HPRT_OK; }
// Diagnostics hook (not part of include/hprt.h): the first batch of the next hprt_render copies the rays that bounce `bounce` queues
// (kind 0: the path segments entering bounce + 1, 1: its shadow rays, 2: its BSDF-sampled light rays) into d_out7 ([7][cap] planes:
// ox oy oz dx dy dz tmax); hprt_debug_captured returns how many.  d_out7 == NULL switches it off.
__attribute__((visibility("default"))) int hprt_debug_capture_rays(HprtScene *s, int bounce, int kind, float *d_out7, size_t cap) {
    if (!s || kind < 0 || kind > 2 || cap > 0x7fffffffull) return HPRT_E_INVALID;
    s->capture.bounce = bounce; s->capture.kind = kind; s->capture.out7 = d_out7; s->capture.cap = cap; s->capture.n = 0;
    return HPRT_OK;
}
static int ApiStreams(HprtScene *s, size_t n, RayStream *rays, HitStream *hits);
// Diagnostics (tools/sort_experiment.py): what consuming rays through a PERMUTED index queue costs.  The n rays of d_rays7 stay where
// they are; d_queue lists them in the order to be traced.  ms[0]: a streaming pass that gathers the rays through the queue into
// [7][n] planes at d_scratch7 (the physical permutation a sort would have to do), ms[1]: k_trace reading the rays through the queue.
__attribute__((visibility("default"))) int hprt_debug_trace_queued(HprtScene *s, size_t n, const float *d_rays7, const uint32_t *d_queue, int anyHit,
                                                                    float *d_scratch7, float ms[2]) try {
    if (!s || !d_rays7 || !d_queue || !d_scratch7 || !ms || n == 0 || n > 0x7ffffff0ull) return SetError(HPRT_E_INVALID, "hprt_debug_trace_queued: bad argument");
    HIP_TRY(hipSetDevice(s->device));
    SceneCall call(s, nullptr);
    RayStream rays; HitStream hits;
    if (int rc = ApiStreams(s, n, &rays, &hits)) return rc;
    DevBuf occ; HIP_TRY(occ.alloc(n));
    hipEvent_t e[3];
    for (auto &x : e) HIP_TRY(hipEventCreate(&x));
    LaunchPackRays(nullptr, d_rays7, (uint32_t)n, rays);
    HIP_TRY(hipEventRecord(e[0], nullptr));
    LaunchCaptureRays(nullptr, d_queue, (uint32_t)n, rays, d_scratch7, (uint32_t)n);
    HIP_TRY(hipEventRecord(e[1], nullptr));
    HitStream none; none.a = nullptr; none.b = nullptr;
    LaunchTrace(nullptr, s->dev, anyHit != 0, false, d_queue, nullptr, (uint32_t)n, (uint32_t)n, rays, anyHit ? none : hits, anyHit ? occ.as<uint8_t>() : nullptr, nullptr, s->workCounter.as<uint32_t>());
    HIP_TRY(hipEventRecord(e[2], nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipEventElapsedTime(&ms[0], e[0], e[1])); HIP_TRY(hipEventElapsedTime(&ms[1], e[1], e[2]));
    for (auto &x : e) (void)hipEventDestroy(x);
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }
__attribute__((visibility("default"))) long long hprt_debug_captured(HprtScene *s) { return s ? (long long)s->capture.n : -1; }
// Measurement hook (not part of include/hprt.h): HBM stream bandwidth of this box, a float4 copy of `bytes` bytes (src and dst far
// larger than the 256 MB Infinity Cache), timed with HIP events over `iters` launches after one warm-up; GB/s count read + write.
__attribute__((visibility("default"))) int hprt_debug_stream_copy(int device, size_t bytes, int iters, double *best_gbs, double *mean_gbs) try {
    if (!best_gbs || !mean_gbs || iters < 1 || iters > 1000 || bytes < (1u << 20) || bytes > (64ull << 30)) return SetError(HPRT_E_INVALID, "hprt_debug_stream_copy: bad argument");
    int dev = 0;
    if (int rc = CheckDevice(device, &dev)) return rc;
    HIP_TRY(hipSetDevice(dev));
    DevBuf a, b;
    HIP_TRY(a.alloc(bytes)); HIP_TRY(b.alloc(bytes));
    HIP_TRY(hipMemset(a.p, 1, bytes)); HIP_TRY(hipMemset(b.p, 2, bytes));
    const size_t n = bytes / 16;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    LaunchStreamCopy(nullptr, a.as<float4>(), b.as<float4>(), n);
    HIP_TRY(hipDeviceSynchronize());
    double best = 0, sum = 0;
    for (int i = 0; i < iters; ++i) {
        HIP_TRY(hipEventRecord(e0, nullptr));
        LaunchStreamCopy(nullptr, a.as<float4>(), b.as<float4>(), n);
        HIP_TRY(hipEventRecord(e1, nullptr));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        const double g = 2.0 * (double)(n * 16) / ((double)ms * 1e-3) / 1e9;
        best = std::max(best, g); sum += g;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *best_gbs = best; *mean_gbs = sum / iters;
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }
// Diagnostics hook (not part of include/hprt.h): rate at which the device serves dependent per-lane gathers of 64-byte records
// over a BVH-like pick from a table of 2^log2_records records (kernels.hip, k_gather_probe) — the measured ceiling bench.py holds
// k_trace's record rate against.  Out: records per second (1e9), best and mean of `launches` timed launches.
__attribute__((visibility("default"))) int hprt_debug_gather_probe(int device, int log2_records, int iters_per_lane, int launches, double *best_grecords_s, double *mean_grecords_s) try {
    if (!best_grecords_s || !mean_grecords_s || log2_records < 8 || log2_records > 26 || iters_per_lane < 1 || iters_per_lane > 100000 || launches < 1 || launches > 100)
        return SetError(HPRT_E_INVALID, "hprt_debug_gather_probe: bad argument");
    int dev = 0;
    if (int rc = CheckDevice(device, &dev)) return rc;
    HIP_TRY(hipSetDevice(dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    const uint32_t blocks = (uint32_t)prop.multiProcessorCount * 6u * 4u;
    DevBuf table, sink;
    const size_t bytes = (size_t)64 << log2_records;
    HIP_TRY(table.alloc(bytes)); HIP_TRY(sink.alloc(16));
    HIP_TRY(hipMemset(table.p, 0x5b, bytes));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    LaunchGatherProbe(nullptr, table.as<uint4>(), (uint32_t)log2_records, std::max(1, iters_per_lane / 10), blocks, sink.as<uint32_t>());
    HIP_TRY(hipDeviceSynchronize());
    double best = 0, sum = 0;
    for (int i = 0; i < launches; ++i) {
        HIP_TRY(hipEventRecord(e0, nullptr));
        LaunchGatherProbe(nullptr, table.as<uint4>(), (uint32_t)log2_records, iters_per_lane, blocks, sink.as<uint32_t>());
        HIP_TRY(hipEventRecord(e1, nullptr));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        const double g = (double)blocks * 256.0 * (double)iters_per_lane / ((double)ms * 1e-3) / 1e9;
        best = std::max(best, g); sum += g;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *best_grecords_s = best; *mean_grecords_s = sum / launches;
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }
__attribute__((visibility("default"))) int hprt_debug_voxel_points(float out[640]) { if (!out) return HPRT_E_INVALID; VoxelSamplePoints(out); return HPRT_OK; }

// Diagnostics hook (not part of include/hprt.h): the restated libm functions of hprt_math.h evaluated ON THE DEVICE over host
// arrays, so that a test can hold them against the oracle's (which equal glibc's, tests/test_oracle_pins.py) directly rather
// than through renders.  fn 0: sinf / cosf of x -> out0 / out1; 1: acosf(x) -> out0; 2: atan2f(y, x) -> out0; 3: logf(x) -> out0;
// 4: double sin / cos of (double)x -> out0 / out1.  Outputs are doubles (a float result converts exactly).
__global__ void k_debug_math(int fn, const float *x, const float *y, size_t n, double *o0, double *o1) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = 0, b = 0;
    if (fn == 0) { float s, c; det_sincosf(x[i], &s, &c); a = (double)s; b = (double)c; }
    else if (fn == 1) a = (double)det_acosf(x[i]);
    else if (fn == 2) a = (double)det_atan2f(y[i], x[i]);
    else if (fn == 3) a = (double)det_logf(x[i]);
    else det_sincos_glibc_d((double)x[i], &a, &b);
    o0[i] = a; o1[i] = b;
}
__attribute__((visibility("default"))) int hprt_debug_device_math(int device, int fn, const float *x, const float *y, size_t n, double *out0, double *out1) try {
    if (fn < 0 || fn > 4 || !x || !y || !out0 || !out1 || n == 0 || n > ((size_t)1 << 28)) return SetError(HPRT_E_INVALID, "hprt_debug_device_math: bad argument");
    HIP_TRY(hipSetDevice(device));
    float *dx = nullptr, *dy = nullptr; double *d0 = nullptr, *d1 = nullptr;
    auto release = [&]() { (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(d0); (void)hipFree(d1); };
    hipError_t e = hipMalloc((void **)&dx, n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&dy, n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d0, n * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d1, n * 8);
    if (e == hipSuccess) e = hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dy, y, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        k_debug_math<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>(fn, dx, dy, n, d0, d1);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out0, d0, n * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out1, d1, n * 8, hipMemcpyDeviceToHost);
    release();
    if (e != hipSuccess) return SetError(HPRT_E_DEVICE, std::string("hprt_debug_device_math: ") + hipGetErrorString(e));
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }

int hprt_scene_create_from_model(const HprtModel *m, const HprtBvh *b, int device, HprtScene **out) try {
    if (!m || !b || !out) return SetError(HPRT_E_INVALID, "hprt_scene_create_from_model: null argument");
    const SceneModel &sm = m->sc;
    std::vector<HprtShapeDesc> shapes(sm.shapes.size());
    for (size_t i = 0; i < sm.shapes.size(); ++i) {
        const ShapeDesc &s = sm.shapes[i];
        HprtShapeDesc &o = shapes[i];
        memset(&o, 0, sizeof(o));
        o.kind = s.kind; o.material = s.material; o.area_light = s.areaLight;
        o.reverse_orientation = s.reverseOrientation; o.transform_swaps_handedness = s.transformSwapsHandedness;
        if (s.kind == kTriangleMesh) {
            o.n_tris = s.mesh.nTris(); o.n_verts = s.mesh.nVerts();
            o.indices = s.mesh.indices.data(); o.P = s.mesh.P.data();
            o.N = s.mesh.N.empty() ? nullptr : s.mesh.N.data();
            o.UV = s.mesh.UV.empty() ? nullptr : s.mesh.UV.data();
            o.S = s.mesh.S.empty() ? nullptr : s.mesh.S.data();
        } else {
            memcpy(o.object_to_world, s.sphere.objectToWorld.m, 64); memcpy(o.world_to_object, s.sphere.worldToObject.m, 64);
            o.radius = s.sphere.radius; o.z_min = s.sphere.zMin; o.z_max = s.sphere.zMax;
            o.theta_min = s.sphere.thetaMin; o.theta_max = s.sphere.thetaMax; o.phi_max = s.sphere.phiMax;
        }
    }
    std::vector<HprtMaterialDesc> mats(sm.materials.size());
    for (size_t i = 0; i < mats.size(); ++i) {
        const MaterialDesc &s = sm.materials[i];
        mats[i].type = s.type; memcpy(mats[i].Kd, s.Kd, 12); mats[i].sigma = s.sigma; memcpy(mats[i].Ks, s.Ks, 12);
        mats[i].roughness = s.roughness; mats[i].remap_roughness = s.remapRoughness;
        mats[i].kd_texture = s.KdTex; mats[i].ks_texture = s.KsTex; mats[i].opacity_texture = s.opacityTex;
        memcpy(mats[i].Kr, s.Kr, 12); memcpy(mats[i].Kt, s.Kt, 12); memcpy(mats[i].opacity, s.opacity, 12); mats[i].eta = s.eta;
    }
    std::vector<std::vector<HprtTextureLevel>> texLevels(sm.textures.size());
    std::vector<HprtTextureDesc> textures(sm.textures.size());
    for (size_t i = 0; i < textures.size(); ++i) {
        const TextureDesc &t = sm.textures[i];
        for (const MipLevel &l : t.levels) texLevels[i].push_back(HprtTextureLevel{l.w, l.h, l.rgb.data()});
        textures[i].levels = texLevels[i].data(); textures[i].n_levels = (uint32_t)texLevels[i].size();
        textures[i].trilinear = t.trilinear; textures[i].max_anisotropy = t.maxAniso; textures[i].wrap = t.wrap;
        textures[i].su = t.su; textures[i].sv = t.sv; textures[i].du = t.du; textures[i].dv = t.dv; textures[i].weight_lut = t.weightLut;
    }
    std::vector<HprtLightDesc> lights(sm.lights.size());
    for (size_t i = 0; i < lights.size(); ++i) {
        const LightDesc &s = sm.lights[i];
        lights[i].type = s.type; memcpy(lights[i].pos, s.pos, 12); memcpy(lights[i].I, s.I, 12); lights[i].shape = s.shape; lights[i].two_sided = s.twoSided;
        lights[i].texture = s.texture; memcpy(lights[i].light_to_world, &s.lightToWorld, 64); memcpy(lights[i].world_to_light, &s.worldToLight, 64);
    }
    if (b->objects.size() != sm.nObjects) return SetError(HPRT_E_INVALID, "the BVH was built for another model (object count differs)");
    // object definitions: their shapes are contiguous (no nesting of definitions, core/api.cpp:1755-1756)
    std::vector<HprtObjectDesc> objects(sm.nObjects);
    for (uint32_t k = 0; k < sm.nObjects; ++k) {
        HprtObjectDesc &o = objects[k];
        memset(&o, 0, sizeof(o));
        bool any = false;
        for (size_t i = 0; i < sm.shapes.size(); ++i)
            if (sm.shapes[i].object == (int32_t)k) { if (!any) { o.first_shape = (uint32_t)i; any = true; } o.n_shapes = (uint32_t)i + 1u - o.first_shape; }
        const BvhTree &t = b->objects[k];
        o.nodes = t.nodes.data(); o.n_nodes = (uint32_t)t.nodes.size(); o.prim_order = t.primOrder.data(); o.n_prims = (uint32_t)t.primOrder.size();
    }
    std::vector<HprtInstanceDesc> instances(sm.instances.size());
    for (size_t i = 0; i < instances.size(); ++i) {
        instances[i].object = sm.instances[i].object;
        memcpy(instances[i].instance_to_world, sm.instances[i].instanceToWorld.m, 64);
        memcpy(instances[i].world_to_instance, sm.instances[i].worldToInstance.m, 64);
    }
    std::vector<HprtTopItem> top(sm.top.size());
    for (size_t i = 0; i < top.size(); ++i) { top[i].kind = sm.top[i].kind; top[i].index = sm.top[i].index; }
    HprtSceneDesc d;
    memset(&d, 0, sizeof(d));
    d.textures = textures.data(); d.n_textures = (uint32_t)textures.size();
    d.objects = objects.data(); d.n_objects = (uint32_t)objects.size();
    d.instances = instances.data(); d.n_instances = (uint32_t)instances.size();
    static const HprtTopItem kNoItems[1] = {{0, 0u}};
    d.top = top.empty() ? kNoItems : top.data(); d.n_top = (uint32_t)top.size();
    d.nodes = b->tree.nodes.data(); d.n_nodes = (uint32_t)b->tree.nodes.size();
    d.prim_order = b->tree.primOrder.data(); d.n_prims = (uint32_t)b->tree.primOrder.size();
    d.shapes = shapes.data(); d.n_shapes = (uint32_t)shapes.size();
    d.materials = mats.data(); d.n_materials = (uint32_t)mats.size();
    d.lights = lights.data(); d.n_lights = (uint32_t)lights.size();
    // a single light always gets the uniform distribution (core/lightdistrib.cpp:50-52)
    d.light_strategy = sm.lights.size() <= 1 ? 0 : sm.opt.lightStrategy;
    return hprt_scene_create(&d, device, out);
} catch (...) { return hprt::HandleException(); }

void hprt_scene_destroy(HprtScene *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    delete s;
}

// ---------------------------------------------------------------------------
// batched Aggregate calls
// ---------------------------------------------------------------------------
static void ReadCounters(HprtScene *s, bool anyHit, uint64_t out[4]) {
    DevCounters c;
    (void)hipMemcpy(&c, s->counters.p, sizeof(c), hipMemcpyDeviceToHost);
    if (!anyHit) { out[0] = c.nodesFetched; out[1] = c.nodesEntered; out[2] = c.triTests; out[3] = c.sphereTests; }
    else { out[0] = c.nodesFetchedP; out[1] = c.nodesEnteredP; out[2] = c.triTestsP; out[3] = c.sphereTestsP; }
}
// Stream copies for the plane-layout entry points ([7][n] rays in; t, prim, [3][n] barycentrics out)
static int ApiStreams(HprtScene *s, size_t n, RayStream *rays, HitStream *hits) {
    HIP_TRY(s->apiRays.alloc(32 * n + 256));
    HIP_TRY(s->apiHits.alloc(24 * n + 256));
    rays->a = s->apiRays.as<float4>(); rays->b = rays->a + n;
    hits->a = s->apiHits.as<float4>(); hits->b = (float2 *)(hits->a + n);
    return HPRT_OK;
}

int hprt_intersect_device(HprtScene *s, size_t n, const float *d_rays7, float *d_t, int32_t *d_prim, float *d_bary3, void *stream) try {
    if (!s || (n && (!d_rays7 || !d_t || !d_prim))) return SetError(HPRT_E_INVALID, "hprt_intersect_device: null argument");
    if (n > 0x7ffffff0ull) return SetError(HPRT_E_INVALID, "too many rays in one call");
    if (n == 0) return HPRT_OK;
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    SceneCall call(s, st);
    RayStream rays; HitStream hits;
    int rc = ApiStreams(s, n, &rays, &hits);
    if (rc != HPRT_OK) return rc;
    LaunchPackRays(st, d_rays7, (uint32_t)n, rays);
    LaunchTrace(st, s->dev, false, false, nullptr, nullptr, (uint32_t)n, (uint32_t)n, rays, hits, nullptr, nullptr, s->workCounter.as<uint32_t>());
    LaunchUnpackHits(st, hits, (uint32_t)n, d_t, d_prim, d_bary3);
    call.leave_async();
    HIP_TRY(hipGetLastError());
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }
int hprt_occluded_device(HprtScene *s, size_t n, const float *d_rays7, uint8_t *d_occ, void *stream) try {
    if (!s || (n && (!d_rays7 || !d_occ))) return SetError(HPRT_E_INVALID, "hprt_occluded_device: null argument");
    if (n > 0x7ffffff0ull) return SetError(HPRT_E_INVALID, "too many rays in one call");
    if (n == 0) return HPRT_OK;
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    SceneCall call(s, st);
    RayStream rays; HitStream hits;
    int rc = ApiStreams(s, n, &rays, &hits);
    if (rc != HPRT_OK) return rc;
    LaunchPackRays(st, d_rays7, (uint32_t)n, rays);
    HitStream none; none.a = nullptr; none.b = nullptr;
    LaunchTrace(st, s->dev, true, false, nullptr, nullptr, (uint32_t)n, (uint32_t)n, rays, none, d_occ, nullptr, s->workCounter.as<uint32_t>());
    call.leave_async();
    HIP_TRY(hipGetLastError());
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }

static int TraceHost(HprtScene *s, bool anyHit, size_t n, const float *o, const float *d, const float *tmax, float *t_out,
                     int32_t *prim_out, int32_t *inst_out, float *bary_out, uint8_t *occ_out, uint64_t counters[4]) {
    if (!s || (n && (!o || !d || !tmax))) return SetError(HPRT_E_INVALID, "trace: null argument");
    if (n > 0x7ffffff0ull) return SetError(HPRT_E_INVALID, "too many rays in one call");
    if (n == 0) { if (counters) memset(counters, 0, 32); return HPRT_OK; }
    HIP_TRY(hipSetDevice(s->device));
    SceneCall call(s, nullptr);      // (blocking call on the null stream: ends in hipDeviceSynchronize)
    std::vector<float4> ra(n), rb(n);
    for (size_t i = 0; i < n; ++i) {
        ra[i] = make_float4(o[3 * i], o[3 * i + 1], o[3 * i + 2], tmax[i]);
        rb[i] = make_float4(d[3 * i], d[3 * i + 1], d[3 * i + 2], 0.f);
    }
    RayStream rays; HitStream hits;
    int rc = ApiStreams(s, n, &rays, &hits);
    if (rc != HPRT_OK) return rc;
    HIP_TRY(hipMemcpy(rays.a, ra.data(), 16 * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(rays.b, rb.data(), 16 * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(s->counters.p, 0, sizeof(DevCounters)));
    const bool count = counters != nullptr;
    if (!anyHit) {
        LaunchTrace(nullptr, s->dev, false, count, nullptr, nullptr, (uint32_t)n, (uint32_t)n, rays, hits, nullptr, s->counters.as<DevCounters>(), s->workCounter.as<uint32_t>());
        HIP_TRY(hipGetLastError()); HIP_TRY(hipDeviceSynchronize());
        std::vector<float4> ha(n); std::vector<float2> hb(n);
        HIP_TRY(hipMemcpy(ha.data(), hits.a, 16 * n, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(hb.data(), hits.b, 8 * n, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) {
            if (t_out) t_out[i] = ha[i].x;
            if (prim_out) { int32_t w; memcpy(&w, &ha[i].y, 4); prim_out[i] = hit_prim(w); }
            if (inst_out) memcpy(&inst_out[i], &hb[i].y, 4);
            if (bary_out) { bary_out[3 * i] = ha[i].z; bary_out[3 * i + 1] = ha[i].w; bary_out[3 * i + 2] = hb[i].x; }
        }
    } else {
        DevBuf outOcc;
        HIP_TRY(outOcc.alloc(n));
        HitStream none; none.a = nullptr; none.b = nullptr;
        LaunchTrace(nullptr, s->dev, true, count, nullptr, nullptr, (uint32_t)n, (uint32_t)n, rays, none, outOcc.as<uint8_t>(), s->counters.as<DevCounters>(), s->workCounter.as<uint32_t>());
        HIP_TRY(hipGetLastError()); HIP_TRY(hipDeviceSynchronize());
        if (occ_out) HIP_TRY(hipMemcpy(occ_out, outOcc.p, n, hipMemcpyDeviceToHost));
    }
    if (counters) ReadCounters(s, anyHit, counters);
    return HPRT_OK;
}
int hprt_intersect(HprtScene *s, size_t n, const float *o, const float *d, const float *tmax, float *t_out, int32_t *prim_out,
                   float *bary_out, uint64_t counters[4]) try {
    return TraceHost(s, false, n, o, d, tmax, t_out, prim_out, nullptr, bary_out, nullptr, counters);
} catch (...) { return hprt::HandleException(); }
int hprt_intersect_instanced(HprtScene *s, size_t n, const float *o, const float *d, const float *tmax, float *t_out, int32_t *prim_out,
                             int32_t *inst_out, float *bary_out, uint64_t counters[4]) try {
    return TraceHost(s, false, n, o, d, tmax, t_out, prim_out, inst_out, bary_out, nullptr, counters);
} catch (...) { return hprt::HandleException(); }
int hprt_occluded(HprtScene *s, size_t n, const float *o, const float *d, const float *tmax, uint8_t *occ, uint64_t counters[4]) try {
    return TraceHost(s, true, n, o, d, tmax, nullptr, nullptr, nullptr, nullptr, occ, counters);
} catch (...) { return hprt::HandleException(); }

// ---------------------------------------------------------------------------
// Render
// ---------------------------------------------------------------------------
namespace {

struct BatchTimers { double extendMs = 0, occludedMs = 0; uint64_t extendLaunches = 0, occludedLaunches = 0, extendRays = 0, occludedRays = 0; };

// Runs the bounce loop for one batch of nSlots freshly generated paths.
// pixelStats (or null): [6][nPix] per-pixel counters of the local pixels, fed from the per-ray counts of every trace
//
// Per bounce b:   trace(path b) -> bin -> shade x3 -> [counts to the host] -> trace(shadow b) | trace(MIS b) | trace(path b+1)
//                 -> resolve(b) -> bin(b+1) ...
// The three traces that follow a shading pass depend on nothing but that pass.  Running them on three HIP streams (so that
// each fills the tails of the others' persistent kernels) was built and measured in round 2: SLOWER — atrium 1024 spp
// 1064 -> 1119 ms, living room 455 -> 478 ms, killeroo-simple 90.6 -> 91.8 ms (every persistent kernel is sized to fill the
// machine, so the second and third only get wave slots as the first one's blocks drain, and the cross-stream waits add
// bubbles) — and removed again: every kernel runs on the caller's stream, where HIP-event times are exclusive.
int RunBatch(HprtScene *s, hipStream_t st, const RenderParams &rp, const Workspace &w, const QueueSet &qa, const QueueSet &qb,
             const BinSet &bins, uint32_t s0, uint32_t nSlots, bool count, EventTimer &ev, BatchTimers *bt, HprtRenderStats *stats,
             uint32_t *pixelStats = nullptr, const IrregularSink *irregular = nullptr) {
    uint4 *rayStats = pixelStats ? s->rayStats.as<uint4>() : nullptr;
    uint32_t *wcPath = s->workCounter.as<uint32_t>();
    LaunchGenerate(st, s->dev, rp, w.path[0], s0, nSlots, irregular);
    const uint32_t *activeQ = nullptr; uint32_t active = nSlots;
    QueueSet q[2] = {qa, qb};
    DevCounters *ctr = s->counters.as<DevCounters>();
    std::vector<std::pair<hipEvent_t, hipEvent_t>> evExt, evOcc;
    auto tracePath = [&](int bounce) -> int {      // closest hits of the path segments entering bounce `bounce`
        const PathStream &in = w.path[bounce & 1];
        hipEvent_t e0 = ev.get(), e1 = ev.get();
        HIP_TRY(hipEventRecord(e0, st));
        LaunchTrace(st, s->dev, false, count, activeQ, nullptr, active, active, in.ray, w.hit, nullptr, ctr, wcPath, rayStats);
        if (pixelStats) LaunchPixelStats(st, rayStats, bounce == 0 ? nullptr : in.beta, activeQ, nullptr, active, active, rp.nPix, false, pixelStats);
        HIP_TRY(hipEventRecord(e1, st));
        evExt.push_back({e0, e1}); bt->extendRays += active; ++bt->extendLaunches;
        stats->rays += active;
        return HPRT_OK;
    };
    if (active > 0) { int rc = tracePath(0); if (rc != HPRT_OK) return rc; }
    for (int bounce = 0; active > 0; ++bounce) {
        const QueueSet &cur = q[bounce & 1];
        const PathStream &in = w.path[bounce & 1], &out = w.path[(bounce + 1) & 1];
        HIP_TRY(hipMemsetAsync(cur.nextCount, 0, 256 * sizeof(uint32_t), st));   // the four counters, 64 words apart
        HIP_TRY(hipMemsetAsync(bins.count, 0, 8 * BIN_STRIDE * sizeof(uint32_t), st));
        LaunchBin(st, s->dev, in, w.hit, activeQ, nullptr, active, active, rp.maxDepth, bounce, bins, w.Lfinal);
        HIP_TRY(hipMemcpyAsync(bins.count + 5 * BIN_STRIDE, bins.count + 2 * BIN_STRIDE, sizeof(uint32_t), hipMemcpyDeviceToDevice, st));   // bin 2 before deferrals
        // one launch per material bin; grids are sized for the upper bound, surplus blocks exit on the bin's count
        // (the specialised variants first: they may hand vertices over to the generic one; bin 3: vertices on image-textured materials)
        static const int order[5] = {BIN_MATTE, BIN_PLASTIC, BIN_SUBSTRATE, BIN_GENERIC, BIN_TEXTURED};
        for (int k = 0; k < (s->dev.textures ? 5 : 4); ++k) {
            const int mode = order[k];
            if (mode == (int)BIN_SUBSTRATE && !s->hasSubstrateBin) continue;
            LaunchShade(st, mode, s->dev, rp, in, w.hit, active, s0, out, w.vs, cur, bins, w.Lfinal, bounce == 0);
        }
        HIP_TRY(hipMemcpyAsync(s->hostCounts + 4096, cur.nextCount, 256 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        if (s->dev.voxSlot) {      // on-demand SpatialLightDistribution: vertices whose voxel had no distribution yet wait in the retry lists
            HIP_TRY(hipMemcpyAsync(s->hostCounts + 16, bins.count + 6 * BIN_STRIDE, 2 * BIN_STRIDE * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(s->hostCounts + 15, s->dev.voxRequestCount, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            const uint32_t nRetry[2] = {s->hostCounts[16], s->hostCounts[16 + BIN_STRIDE]}, nReq = s->hostCounts[15];
            if (nRetry[0] + nRetry[1] > 0) {
                if ((uint64_t)s->voxRowsUsed + nReq > s->voxRows)
                    return SetError(HPRT_E_UNSUPPORTED, "spatial light distribution: the paths touch more voxels than the row pool holds (" + std::to_string(s->voxRows) +
                                                        " rows of " + std::to_string(s->dev.nLights) + " lights; HPRT_VOXEL_POOL_MB raises it, or use \"power\" / \"uniform\")");
                // SpatialLightDistribution::ComputeDistribution for the voxels just asked for, then the waiting vertices again
                LaunchVoxelFill(st, s->dev, s->voxRi.as<float>(), nReq, s->voxRowsUsed, s->voxFunc.as<float>(), s->voxCdf.as<float>(), s->voxFuncInt.as<float>());
                s->voxRowsUsed += nReq;
                HIP_TRY(hipMemsetAsync(s->dev.voxRequestCount, 0, sizeof(uint32_t), st));
                for (int r = 0; r < 2; ++r)
                    if (nRetry[r]) LaunchShade(st, 2 + r, s->dev, rp, in, w.hit, nRetry[r], s0, out, w.vs, cur, bins, w.Lfinal, bounce == 0, true);
                HIP_TRY(hipMemcpyAsync(s->hostCounts + 4096, cur.nextCount, 256 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            }
        }
        HIP_TRY(hipStreamSynchronize(st));
        const uint32_t nNext = s->hostCounts[4096], nShadow = s->hostCounts[4096 + 64], nMis = s->hostCounts[4096 + 128], nResolve = s->hostCounts[4096 + 192];
        if (s->capture.out7 && s->capture.bounce == bounce && s0 == 0) {      // diagnostics (hprt_debug_capture_rays)
            const uint32_t *cq = s->capture.kind == 0 ? cur.next : s->capture.kind == 1 ? cur.shadow : cur.mis;
            const uint32_t cn = s->capture.kind == 0 ? nNext : s->capture.kind == 1 ? nShadow : nMis;
            const RayStream &cs = s->capture.kind == 0 ? out.ray : s->capture.kind == 1 ? w.vs.shadow : w.vs.mis;
            LaunchCaptureRays(st, cq, cn, cs, s->capture.out7, (uint32_t)s->capture.cap);
            s->capture.n = std::min<size_t>(cn, s->capture.cap);
        }
        if (nShadow) {
            hipEvent_t a = ev.get(), b = ev.get();
            HIP_TRY(hipEventRecord(a, st));
            HitStream none; none.a = nullptr; none.b = nullptr;
            LaunchTrace(st, s->dev, true, count, cur.shadow, nullptr, nShadow, nShadow, w.vs.shadow, none, w.vs.occluded, ctr, wcPath, rayStats);
            if (pixelStats) LaunchPixelStats(st, rayStats, w.vs.pendBeta, cur.shadow, nullptr, nShadow, nShadow, rp.nPix, true, pixelStats);
            HIP_TRY(hipEventRecord(b, st));
            evOcc.push_back({a, b}); bt->occludedRays += nShadow; ++bt->occludedLaunches;
            stats->shadow_rays += nShadow;
        }
        if (nMis) {
            hipEvent_t a = ev.get(), b = ev.get();
            HIP_TRY(hipEventRecord(a, st));
            LaunchTrace(st, s->dev, false, count, cur.mis, nullptr, nMis, nMis, w.vs.mis, w.vs.misHit, nullptr, ctr, wcPath, rayStats);
            if (pixelStats) LaunchPixelStats(st, rayStats, w.vs.pendBeta, cur.mis, nullptr, nMis, nMis, rp.nPix, false, pixelStats);
            HIP_TRY(hipEventRecord(b, st));
            evExt.push_back({a, b}); bt->extendRays += nMis; ++bt->extendLaunches;
            stats->rays += nMis;
        }
        activeQ = cur.next; active = nNext;
        // (maxDepth is bounded by CheckDepth, and no path outlives bounce maxDepth)
        if (active > 0 && bounce > rp.maxDepth) return SetError(HPRT_E_DEVICE, "internal error: paths are still active beyond maxdepth");
        if (nResolve) LaunchResolve(st, s->dev, w.vs, out.L, w.Lfinal, cur.resolve, cur.resolveCount, nResolve);
        if (active > 0) { int rc = tracePath(bounce + 1); if (rc != HPRT_OK) return rc; }
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    for (auto &p : evExt) { float ms = 0; if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) bt->extendMs += ms; }
    for (auto &p : evOcc) { float ms = 0; if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) bt->occludedMs += ms; }
    ev.reset();
    return HPRT_OK;
}

int EnsureWorkspace(HprtScene *s, size_t nSlots, Workspace *ps, QueueSet *qa, QueueSet *qb, BinSet *bins) {
    HIP_TRY(s->planes.alloc(PlaneBytes(nSlots)));
    CarvePlanes(s->planes.as<char>(), nSlots, ps);
    HIP_TRY(s->queues.alloc(14 * nSlots * sizeof(uint32_t) + 4096));
    HIP_TRY(s->queueCounts.alloc(2048 * sizeof(uint32_t)));
    uint32_t *qbase = s->queues.as<uint32_t>(), *cbase = s->queueCounts.as<uint32_t>();
    QueueSet *qs[2] = {qa, qb};
    for (int k = 0; k < 2; ++k) {
        qs[k]->next = qbase + (4 * k + 0) * nSlots; qs[k]->shadow = qbase + (4 * k + 1) * nSlots;
        qs[k]->mis = qbase + (4 * k + 2) * nSlots; qs[k]->resolve = qbase + (4 * k + 3) * nSlots;
        // one counter per 256-byte line: atomics of different queues do not serialise on a shared line
        qs[k]->nextCount = cbase + 256 * k; qs[k]->shadowCount = cbase + 256 * k + 64; qs[k]->misCount = cbase + 256 * k + 128; qs[k]->resolveCount = cbase + 256 * k + 192;
    }
    for (int k = 0; k < (int)N_BINS; ++k) bins->q[k] = qbase + (8 + k) * nSlots;
    bins->aux = qbase + 13 * nSlots;
    bins->count = cbase + 512;
    bins->retry[0] = bins->retry[1] = nullptr;
    if (s->dev.voxSlot) {      // (on-demand voxel tables only)
        HIP_TRY(s->retryQueues.alloc(2 * nSlots * sizeof(uint2) + 256));
        bins->retry[0] = s->retryQueues.as<uint2>(); bins->retry[1] = bins->retry[0] + nSlots;
    }
    return HPRT_OK;
}

// Test hook: garbage in everything a render treats as scratch.  A render whose film changes under it has consumed a word
// it never wrote (a fresh allocation on a busy card holds other processes' data, not zeros).
int PoisonWorkspace(HprtScene *s) {
    if (s->poisonByte < 0) return HPRT_OK;
    hprt::DevBuf *bufs[] = {&s->planes, &s->queues, &s->Lall, &s->deepStack, &s->rayStats, &s->irregular};
    for (hprt::DevBuf *b : bufs) if (b->p && b->bytes) HIP_TRY(hipMemset(b->p, s->poisonByte, b->bytes));
    HIP_TRY(hipDeviceSynchronize());
    return HPRT_OK;
}

// Samples per pixel of one wavefront batch (see the comment at its use in hprt_render)
int ChooseBatch(HprtScene *s, int32_t sppChunk, uint32_t nPix, uint32_t spp, uint32_t *out) {
    uint32_t chunk;
    if (sppChunk > 0) chunk = (uint32_t)sppChunk;
    else {
        size_t freeB = 0, totalB = 0;
        HIP_TRY(hipMemGetInfo(&freeB, &totalB));
        freeB += s->planes.bytes + s->queues.bytes;                 // this scene's previous workspace is reused or released
        const size_t perPath = kPlaneBytesPerSlot + 14 * sizeof(uint32_t);
        static const size_t capM = [] { const char *e = getenv("HPRT_BATCH_MPATHS"); return e ? (size_t)atoi(e) : (size_t)256; }();
        const size_t budget = std::min<size_t>(capM << 20, std::max<size_t>(freeB / 2 / perPath, 1ull << 20));
        chunk = std::max<uint32_t>(1u, (uint32_t)(budget / std::max<uint32_t>(nPix, 1u)));
        if ((uint64_t)chunk * nPix > 0x7fffffffull) chunk = (uint32_t)(0x7fffffffull / nPix);
        chunk = std::max(1u, std::min(chunk, spp));
        const uint32_t nBatches = (spp + chunk - 1) / chunk;
        chunk = (spp + nBatches - 1) / nBatches;                    // equal batches
    }
    chunk = std::min(chunk, spp);
    if ((uint64_t)chunk * nPix > 0x7fffffffull) chunk = (uint32_t)(0x7fffffffull / nPix);
    *out = chunk;
    return HPRT_OK;
}

// Host-side grouping of the irregular samples into per-destination lists, in the exact
// order the reference's tile loop would have added them (core/integrator.cpp:267-333).
struct ExtraEntry { uint32_t dest; uint32_t srcTile; uint32_t srcPos; uint32_t sample; uint32_t srcPix; uint8_t pre; };

}  // namespace

int hprt_render(HprtScene *s, const HprtRenderDesc *desc, float *d_film_xyzw, void *stream, HprtRenderStats *stats) try {
    if (!s || !desc) return SetError(HPRT_E_INVALID, "hprt_render: null argument");
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    SceneCall call(s, st);      // (blocking: every path out of a started render has synchronised `st`, or failed)
    const HprtRenderOptions &o = desc->opt;
    int rc0;
    if (o.spp <= 0 || o.max_depth < 0) return SetError(HPRT_E_INVALID, "spp must be positive and max_depth non-negative");
    if ((rc0 = CheckDepth(o.max_depth)) != HPRT_OK) return rc0;
    FrameSetup f;
    int rc = SetupFrame(o, &f);
    if (rc != HPRT_OK) return rc;
    HprtRenderStats localStats; if (!stats) stats = &localStats;
    memset(stats, 0, sizeof(*stats));
    // ---- local tiles ----
    const int nTiles = f.ntx * f.nty;
    int tb = std::max(0, desc->tile_begin), te = desc->tile_end <= 0 ? nTiles : std::min(desc->tile_end, nTiles), ts = std::max(1, desc->tile_stride);
    f.localIndex.assign((size_t)f.W * f.H, -1);
    f.pixelXY.reserve((size_t)f.W * f.H / (size_t)ts + 256); f.pixelOffset.reserve((size_t)f.W * f.H / (size_t)ts + 256);
    std::vector<uint8_t> tileIsLocal(nTiles, 0);
    for (int t = tb; t < te; t += ts) { tileIsLocal[t] = 1; AddTilePixels(&f, t); }
    const uint32_t nPix = (uint32_t)f.pixelXY.size();
    const uint32_t spp = (uint32_t)o.spp;
    s->filmPixels = (size_t)f.W * f.H;
    s->filmW = f.W; s->filmH = f.H;
    s->nForeignRecords = 0; s->foreignExported = (desc->flags & HPRT_RENDER_EXPORT_FOREIGN) != 0;
    float *film = d_film_xyzw;
    if (!film) { HIP_TRY(s->film.alloc(16 * s->filmPixels)); film = s->film.as<float>(); }
    s->lastFilm = film;
    HIP_TRY(hipMemsetAsync(film, 0, 16 * s->filmPixels, st));
    if (nPix == 0) { HIP_TRY(hipStreamSynchronize(st)); return HPRT_OK; }
    // ---- sizes ----
    const size_t lallBytes = 3ull * spp * nPix * sizeof(float);
    if (lallBytes > (96ull << 30)) return SetError(HPRT_E_UNSUPPORTED, "per-sample radiance store would exceed 96 GiB; render in several tile ranges");
    // Paths per wavefront batch.  Every launch ends with a tail of half-empty waves and every bounce with a host
    // read-back, so batches are as large as memory allows: up to 256 M paths (393 B per path of streams and queues = 100 GB
    // of the 288 GB), less if the device has less free (half of what is free now), and of EQUAL size (1,024 spp of the atrium:
    // two batches of 512 instead of 522 + 502 or, at the old 128 M cap, four of 256: -1 % per frame).  killeroo-simple
    // at 256 spp is one batch of 125 M paths: 6.5 % faster than two batches of 64 M.  (HPRT_BATCH_MPATHS changes the cap.)
    uint32_t chunk = 0;
    if ((rc = ChooseBatch(s, desc->spp_chunk, nPix, spp, &chunk)) != HPRT_OK) return rc;
    const size_t maxSlots = (size_t)chunk * nPix;
    Workspace ps; QueueSet qa, qb; BinSet bins;
    rc = EnsureWorkspace(s, maxSlots, &ps, &qa, &qb, &bins);
    if (rc != HPRT_OK) return rc;
    HIP_TRY(s->Lall.alloc(lallBytes));
    if ((rc = PoisonWorkspace(s)) != HPRT_OK) return rc;
    float *LallR = s->Lall.as<float>(), *LallG = LallR + (size_t)spp * nPix, *LallB = LallG + (size_t)spp * nPix;
    HIP_TRY(upload(s->pixelXY, f.pixelXY)); HIP_TRY(upload(s->pixelOffset, f.pixelOffset));
    RenderParams rp;
    MakeCamera(o, &rp.cam);
    rp.hal.baseScale1 = f.hal.baseScales[1]; rp.hal.baseExp0 = f.hal.baseExponents[0]; rp.hal.sampleStride = f.hal.sampleStride;
    rp.hal.samplePixelCenter = o.sample_pixel_center;
    rp.pixelXY = s->pixelXY.as<uint32_t>(); rp.pixelOffset = s->pixelOffset.as<uint64_t>(); rp.nPix = nPix;
    rp.maxDepth = o.max_depth; rp.rrThreshold = o.rr_threshold;
    rp.invSqrtSpp = 1 / std::sqrt((float)spp);      // ScaleDifferentials' factor, core/integrator.cpp:288-289
    const bool wantPixelStats = (desc->flags & HPRT_RENDER_PIXEL_STATS) != 0;
    const bool count = (desc->flags & HPRT_RENDER_COUNT_WORK) != 0 || wantPixelStats;
    // a counting render traces exactly the reference's rays (its counters are the reference's) unless asked to count what a plain render traces
    rp.cullMis = (!count || (desc->flags & HPRT_RENDER_COUNT_TRACED) != 0) && !(desc->flags & HPRT_RENDER_TRACE_ALL) ? 1 : 0;
    uint32_t *pixelStats = nullptr;
    if (wantPixelStats) {
        HIP_TRY(s->rayStats.alloc(sizeof(uint4) * maxSlots));
        HIP_TRY(s->pixelStatsLocal.alloc(6ull * nPix * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(s->pixelStatsLocal.p, 0, 6ull * nPix * sizeof(uint32_t), st));
        HIP_TRY(s->pixelStatsFilm.alloc(7ull * s->filmPixels * sizeof(uint64_t)));
        HIP_TRY(hipMemsetAsync(s->pixelStatsFilm.p, 0, 7ull * s->filmPixels * sizeof(uint64_t), st));
        pixelStats = s->pixelStatsLocal.as<uint32_t>();
    }
    s->pixelStatsValid = false;
    HIP_TRY(hipMemsetAsync(s->counters.p, 0, sizeof(DevCounters), st));

    auto wall0 = std::chrono::high_resolution_clock::now();
    // ---- film footprint pre-pass ----
    // Irregular film samples: k_generate lists them while it forms the camera samples (capacity: a first guess — a few samples per
    // pixel have a zero Halton offset); if the guess was too small, k_find_irregular recounts with the counted size
    // (HPRT_IRREGULAR_CAP: test hook for that fallback)
    static const uint32_t capHint = [] { const char *e = getenv("HPRT_IRREGULAR_CAP"); return e ? (uint32_t)std::max(1, atoi(e)) : (1u << 24); }();
    uint32_t irrCap = (uint32_t)std::min<uint64_t>((uint64_t)nPix * spp, capHint);
    HIP_TRY(s->irregularCount.alloc(16));
    HIP_TRY(s->irregular.alloc((size_t)irrCap * sizeof(IrregularSample)));
    HIP_TRY(hipMemsetAsync(s->irregularCount.p, 0, 16, st));
    const IrregularSink sink = {f.fg, s->irregularCount.as<uint32_t>(), irrCap, s->irregular.as<IrregularSample>()};
    // ---- batches ----
    EventTimer ev; BatchTimers bt;
    for (uint32_t s0 = 0; s0 < spp; s0 += chunk) {
        const uint32_t c = std::min(chunk, spp - s0), nSlots = c * nPix;
        rc = RunBatch(s, st, rp, ps, qa, qb, bins, s0, nSlots, count, ev, &bt, stats, pixelStats, &sink);
        if (rc != HPRT_OK) return rc;
        LaunchStoreRadiance(st, ps.Lfinal, LallR, LallG, LallB, nPix, s0, nSlots);
    }
    HIP_TRY(hipMemcpyAsync(s->hostCounts + 8, s->irregularCount.p, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    uint32_t nIrr = s->hostCounts[8];
    if (nIrr > irrCap) {      // the list overflowed: once more, standalone, with room for all of them
        if ((uint64_t)nIrr * sizeof(IrregularSample) > (8ull << 30)) return SetError(HPRT_E_UNSUPPORTED, "too many irregular film samples");
        irrCap = nIrr;
        HIP_TRY(s->irregular.alloc((size_t)irrCap * sizeof(IrregularSample)));
        HIP_TRY(hipMemsetAsync(s->irregularCount.p, 0, 16, st));
        LaunchFindIrregular(st, s->dev, rp, f.fg, spp, s->irregularCount.as<uint32_t>(), irrCap, s->irregular.as<IrregularSample>());
        HIP_TRY(hipMemcpyAsync(s->hostCounts + 8, s->irregularCount.p, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (s->hostCounts[8] != nIrr) return SetError(HPRT_E_DEVICE, "internal error: irregular film samples counted differently");
    }
    std::vector<IrregularSample> irr(nIrr);
    if (nIrr) HIP_TRY(hipMemcpy(irr.data(), s->irregular.p, (size_t)nIrr * sizeof(IrregularSample), hipMemcpyDeviceToHost));
    std::vector<ExtraEntry> own, foreign;
    for (const IrregularSample &r : irr) {
        const uint32_t pxy = f.pixelXY[r.pix];
        const int qx = (int)(pxy & 0xffffu), qy = (int)(pxy >> 16);
        const int qtx = (qx - f.fg.sx0) / 16, qty = (qy - f.fg.sy0) / 16;
        const uint32_t qTile = (uint32_t)(qty * f.ntx + qtx);
        const uint32_t qPos = (uint32_t)((qy - (f.fg.sy0 + qty * 16)) * 16 + (qx - (f.fg.sx0 + qtx * 16)));
        for (int y = r.y0; y < r.y1; ++y)
            for (int x = r.x0; x < r.x1; ++x) {
                if (x == qx && y == qy) continue;
                const int dtx = (x - f.fg.sx0) / 16, dty = (y - f.fg.sy0) / 16;
                const uint32_t filmIdx = (uint32_t)((size_t)(y - f.fg.cy0) * f.W + (x - f.fg.cx0));
                ExtraEntry e; e.srcTile = qTile; e.srcPos = qPos; e.sample = r.sample; e.srcPix = r.pix;
                if (dtx == qtx && dty == qty) {
                    e.dest = (uint32_t)f.localIndex[filmIdx];
                    e.pre = (qy < y || (qy == y && qx < x)) ? 1 : 0;
                    own.push_back(e);
                } else { e.dest = filmIdx; e.pre = 0; foreign.push_back(e); }
            }
    }
    std::sort(own.begin(), own.end(), [](const ExtraEntry &a, const ExtraEntry &b) {
        if (a.dest != b.dest) return a.dest < b.dest;
        if (a.srcPos != b.srcPos) return a.srcPos < b.srcPos;   // pre entries have smaller positions than the destination's
        return a.sample < b.sample;
    });
    std::sort(foreign.begin(), foreign.end(), [](const ExtraEntry &a, const ExtraEntry &b) {
        if (a.dest != b.dest) return a.dest < b.dest;
        if (a.srcTile != b.srcTile) return a.srcTile < b.srcTile;
        if (a.srcPos != b.srcPos) return a.srcPos < b.srcPos;
        return a.sample < b.sample;
    });
    FilmExtras ex; memset(&ex, 0, sizeof(ex));
    std::vector<uint32_t> groupDest, groupTile;      // per foreign group (destination film pixel, source tile), sorted by both
    {
        std::vector<uint32_t> begin(nPix + 1, 0), src(own.size()), smp(own.size()); std::vector<uint8_t> pre(own.size());
        for (const ExtraEntry &e : own) ++begin[e.dest + 1];
        for (uint32_t i = 0; i < nPix; ++i) begin[i + 1] += begin[i];
        for (size_t i = 0; i < own.size(); ++i) { src[i] = own[i].srcPix; smp[i] = own[i].sample; pre[i] = own[i].pre; }
        HIP_TRY(upload(s->exOwnBegin, begin)); HIP_TRY(upload(s->exOwnSrc, src)); HIP_TRY(upload(s->exOwnSample, smp)); HIP_TRY(upload(s->exOwnPre, pre));
        ex.ownBegin = s->exOwnBegin.as<uint32_t>(); ex.ownSrcPix = s->exOwnSrc.as<uint32_t>(); ex.ownSample = s->exOwnSample.as<uint32_t>(); ex.ownIsPre = s->exOwnPre.as<uint8_t>();
        std::vector<uint32_t> dest, destBegin, groupBegin, fsrc(foreign.size()), fsmp(foreign.size());
        for (size_t i = 0; i < foreign.size(); ++i) {
            const bool newDest = i == 0 || foreign[i].dest != foreign[i - 1].dest;
            const bool newGroup = newDest || foreign[i].srcTile != foreign[i - 1].srcTile;
            if (newDest) { dest.push_back(foreign[i].dest); destBegin.push_back((uint32_t)groupBegin.size()); }
            if (newGroup) { groupBegin.push_back((uint32_t)i); groupDest.push_back(foreign[i].dest); groupTile.push_back(foreign[i].srcTile); }
            fsrc[i] = foreign[i].srcPix; fsmp[i] = foreign[i].sample;
        }
        destBegin.push_back((uint32_t)groupBegin.size()); groupBegin.push_back((uint32_t)foreign.size());
        HIP_TRY(upload(s->exFDest, dest)); HIP_TRY(upload(s->exFDestBegin, destBegin)); HIP_TRY(upload(s->exFGroupBegin, groupBegin));
        HIP_TRY(upload(s->exFSrc, fsrc)); HIP_TRY(upload(s->exFSample, fsmp));
        ex.nForeignDest = (uint32_t)dest.size();
        ex.foreignDestFilmIndex = s->exFDest.as<uint32_t>(); ex.foreignDestBegin = s->exFDestBegin.as<uint32_t>(); ex.foreignGroupBegin = s->exFGroupBegin.as<uint32_t>();
        ex.foreignSrcPix = s->exFSrc.as<uint32_t>(); ex.foreignSample = s->exFSample.as<uint32_t>();
    }
    LaunchFilmOwn(st, rp, f.fg, LallR, LallG, LallB, spp, ex, film);
    if (desc->flags & HPRT_RENDER_EXPORT_FOREIGN) {
        // cross-tile contributions leave as records: hprt_film_gather merges those of all ranks in source-tile order
        const uint32_t nGroups = (uint32_t)groupDest.size();
        HIP_TRY(upload(s->exGroupDest, groupDest)); HIP_TRY(upload(s->exGroupTile, groupTile));
        HIP_TRY(s->foreignRecords.alloc(std::max<size_t>(1, nGroups) * sizeof(FilmRecord)));
        LaunchFilmForeignExport(st, rp, f.fg, LallR, LallG, LallB, ex, nGroups, s->exGroupDest.as<uint32_t>(), s->exGroupTile.as<uint32_t>(),
                                s->foreignRecords.as<FilmRecord>());
        s->nForeignRecords = nGroups;
    } else
        LaunchFilmForeign(st, rp, f.fg, LallR, LallG, LallB, ex, film);
    if (pixelStats) {
        LaunchPixelStatsToFilm(st, pixelStats, rp.pixelXY, nPix, spp, f.fg.cx0, f.fg.cy0, f.W, s->pixelStatsFilm.as<unsigned long long>());
        s->pixelStatsValid = true;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    stats->render_seconds = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - wall0).count();
    stats->camera_rays = (uint64_t)nPix * spp;
    stats->extend_seconds = bt.extendMs * 1e-3; stats->occluded_seconds = bt.occludedMs * 1e-3;
    stats->extend_launches = bt.extendLaunches; stats->occluded_launches = bt.occludedLaunches;
    stats->extend_rays = bt.extendRays; stats->occluded_rays = bt.occludedRays;
    if (count) {
        DevCounters c;
        HIP_TRY(hipMemcpy(&c, s->counters.p, sizeof(c), hipMemcpyDeviceToHost));
        stats->nodes_fetched = c.nodesFetched; stats->nodes_entered = c.nodesEntered; stats->tri_tests = c.triTests; stats->sphere_tests = c.sphereTests;
        stats->nodes_fetched_p = c.nodesFetchedP; stats->nodes_entered_p = c.nodesEnteredP; stats->tri_tests_p = c.triTestsP; stats->sphere_tests_p = c.sphereTestsP;
    }
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }

// Pays for the coming hprt_render(s, desc, ...) at load time: the wavefront workspace (path streams and queues: ~393 B per path of
// a batch, 105 GB for a 256 M-path batch) and the per-sample radiance store are allocated now, so that the render itself starts
// with its first kernel.  (A fresh process gets 100 GB in under a millisecond, but right after another process released as much the
// driver may spend seconds reclaiming it inside hipMalloc — DESIGN.md §7; a pbrt host renders once and would pay that inside Render().)
int hprt_scene_reserve(HprtScene *s, const HprtRenderDesc *desc) try {
    if (!s || !desc) return SetError(HPRT_E_INVALID, "hprt_scene_reserve: null argument");
    const HprtRenderOptions &o = desc->opt;
    if (o.spp <= 0 || o.max_depth < 0) return SetError(HPRT_E_INVALID, "spp must be positive and max_depth non-negative");
    HIP_TRY(hipSetDevice(s->device));
    SceneCall call(s, nullptr);
    FrameSetup f;
    int rc = SetupFrame(o, &f);
    if (rc != HPRT_OK) return rc;
    const int nTiles = f.ntx * f.nty;
    const int tb = std::max(0, desc->tile_begin), te = desc->tile_end <= 0 ? nTiles : std::min(desc->tile_end, nTiles), ts = std::max(1, desc->tile_stride);
    uint64_t nPix64 = 0;
    for (int t = tb; t < te; t += ts) {
        const int tx = t % f.ntx, ty = t / f.ntx;
        const int x0 = f.fg.sx0 + tx * 16, x1 = std::min(x0 + 16, f.fg.sx1), y0 = f.fg.sy0 + ty * 16, y1 = std::min(y0 + 16, f.fg.sy1);
        nPix64 += (uint64_t)(x1 - x0) * (uint64_t)(y1 - y0);
    }
    if (nPix64 == 0) return HPRT_OK;
    const uint32_t nPix = (uint32_t)nPix64, spp = (uint32_t)o.spp;
    const size_t lallBytes = 3ull * spp * nPix * sizeof(float);
    if (lallBytes > (96ull << 30)) return SetError(HPRT_E_UNSUPPORTED, "per-sample radiance store would exceed 96 GiB; render in several tile ranges");
    uint32_t chunk = 0;
    if ((rc = ChooseBatch(s, desc->spp_chunk, nPix, spp, &chunk)) != HPRT_OK) return rc;
    Workspace ps; QueueSet qa, qb; BinSet bins;
    if ((rc = EnsureWorkspace(s, (size_t)chunk * nPix, &ps, &qa, &qb, &bins)) != HPRT_OK) return rc;
    HIP_TRY(s->Lall.alloc(lallBytes));
    HIP_TRY(s->pixelXY.alloc((size_t)nPix * sizeof(uint32_t))); HIP_TRY(s->pixelOffset.alloc((size_t)nPix * sizeof(uint64_t)));
    HIP_TRY(hipDeviceSynchronize());
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }

// Pixel::stats of the last hprt_render with HPRT_RENDER_PIXEL_STATS (core/film.h:91): 7 values per film pixel
int hprt_pixel_stats_read(HprtScene *s, uint64_t *out7, size_t n_pixels) try {
    if (!s || !out7) return SetError(HPRT_E_INVALID, "hprt_pixel_stats_read: null argument");
    SceneCall call(s, nullptr);
    if (!s->pixelStatsValid) return SetError(HPRT_E_INVALID, "no per-pixel statistics: render with HPRT_RENDER_PIXEL_STATS first");
    if (n_pixels != s->filmPixels) return SetError(HPRT_E_INVALID, "hprt_pixel_stats_read: pixel count differs from the last render's film");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemcpy(out7, s->pixelStatsFilm.p, 7 * n_pixels * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }
int hprt_film_read(HprtScene *s, float *xyzw_out, size_t n_pixels) try {
    if (!s || !xyzw_out) return SetError(HPRT_E_INVALID, "hprt_film_read: null argument");
    SceneCall call(s, nullptr);
    if (!s->film.p || s->lastFilm != s->film.as<float>() || n_pixels != s->filmPixels)
        return SetError(HPRT_E_INVALID, "hprt_film_read: the last render did not write a library-owned film of that size (render with d_film_xyzw == NULL first)");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemcpy(xyzw_out, s->film.p, 16 * n_pixels, hipMemcpyDeviceToHost));
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }

int hprt_sample_radiance(HprtScene *s, const HprtRenderOptions *opt, size_t n, const int32_t *px, const int32_t *py, const int64_t *sample,
                         float *L_out) try {
    if (!s || !opt || (n && (!px || !py || !sample || !L_out))) return SetError(HPRT_E_INVALID, "hprt_sample_radiance: null argument");
    if (n == 0) return HPRT_OK;
    if (n > (1u << 26)) return SetError(HPRT_E_INVALID, "hprt_sample_radiance: too many samples in one call");
    if (opt->max_depth < 0) return SetError(HPRT_E_INVALID, "max_depth must be non-negative");
    if (int rcd = CheckDepth(opt->max_depth)) return rcd;
    HIP_TRY(hipSetDevice(s->device));
    SceneCall call(s, nullptr);
    FrameSetup f;
    int rc = SetupFrame(*opt, &f);
    if (rc != HPRT_OK) return rc;
    std::vector<uint32_t> xy(n); std::vector<uint64_t> off(n);
    for (size_t i = 0; i < n; ++i) {
        if (px[i] < f.fg.sx0 || px[i] >= f.fg.sx1 || py[i] < f.fg.sy0 || py[i] >= f.fg.sy1 || sample[i] < 0)
            return SetError(HPRT_E_INVALID, "hprt_sample_radiance: pixel outside the sample bounds");
        xy[i] = (uint32_t)px[i] | ((uint32_t)py[i] << 16);
        off[i] = (uint64_t)HaltonPixelOffset(f.hal, px[i], py[i]) + (uint64_t)sample[i] * (uint64_t)f.hal.sampleStride;
    }
    Workspace ps; QueueSet qa, qb; BinSet bins;
    rc = EnsureWorkspace(s, n, &ps, &qa, &qb, &bins);
    if (rc != HPRT_OK) return rc;
    HIP_TRY(upload(s->pixelXY, xy)); HIP_TRY(upload(s->pixelOffset, off));
    HIP_TRY(s->Lall.alloc(12 * n));
    if ((rc = PoisonWorkspace(s)) != HPRT_OK) return rc;
    RenderParams rp;
    MakeCamera(*opt, &rp.cam);
    rp.hal.baseScale1 = f.hal.baseScales[1]; rp.hal.baseExp0 = f.hal.baseExponents[0]; rp.hal.sampleStride = f.hal.sampleStride;
    rp.hal.samplePixelCenter = opt->sample_pixel_center;
    rp.pixelXY = s->pixelXY.as<uint32_t>(); rp.pixelOffset = s->pixelOffset.as<uint64_t>(); rp.nPix = (uint32_t)n;
    rp.maxDepth = opt->max_depth; rp.rrThreshold = opt->rr_threshold;
    rp.invSqrtSpp = 1 / std::sqrt((float)std::max(1, opt->spp)); rp.cullMis = 1;
    EventTimer ev; BatchTimers bt; HprtRenderStats stats; memset(&stats, 0, sizeof(stats));
    rc = RunBatch(s, nullptr, rp, ps, qa, qb, bins, 0, (uint32_t)n, false, ev, &bt, &stats);
    if (rc != HPRT_OK) return rc;
    float *LR = s->Lall.as<float>();
    LaunchStoreRadiance(nullptr, ps.Lfinal, LR, LR + n, LR + 2 * n, (uint32_t)n, 0, (uint32_t)n);
    HIP_TRY(hipGetLastError()); HIP_TRY(hipDeviceSynchronize());
    std::vector<float> planes(3 * n);
    HIP_TRY(hipMemcpy(planes.data(), LR, 12 * n, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) { L_out[3 * i] = planes[i]; L_out[3 * i + 1] = planes[n + i]; L_out[3 * i + 2] = planes[2 * n + i]; }
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }

}  // extern "C"
