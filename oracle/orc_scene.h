// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
// In-memory scene + reader for the baked scene container ("HPRTSCN1", layout
// documented in DESIGN.md §Baked scene).  The container holds the post-parse,
// world-space scene exactly as the reference's MakeScene() would hand it to
// BVHAccel (core/api.cpp:1883-1892): shapes in creation order, world-space
// mesh vertices (shapes/triangle.cpp:72-88), materials, lights, camera.
#pragma once
#include <cstdio>
#include <string>
#include <vector>
#include "orc_math.h"

namespace orc {

enum { MAT_MATTE = 0, MAT_PLASTIC = 1, MAT_MIRROR = 2, MAT_SUBSTRATE = 3, MAT_METAL = 4, MAT_GLASS = 5, MAT_UBER = 6 };      // glass: Kd = Kt, Ks = Kr, roughness = eta
// mirror: Kr travels in Ks; substrate: roughness = uroughness, sigma = vroughness; metal: Kd = eta, Ks = k, roughness / sigma likewise
enum { LIGHT_POINT = 0, LIGHT_DISTANT = 1, LIGHT_AREA = 2, LIGHT_INFINITE = 3 };
enum { SHAPE_MESH = 0, SHAPE_SPHERE = 1 };

struct Material {
    int type;
    Float Kd[3]; Float sigma; Float Ks[3]; Float roughness; int remap;
    int KdTex = -1, KsTex = -1;       // Scene::textures index when the parameter is an ImageTexture
    // uber (materials/uber.cpp; roughness = uroughness, sigma = vroughness): the lobes the other materials do not have
    Float Kr[3] = {0, 0, 0}, Kt[3] = {0, 0, 0}, opacity[3] = {1, 1, 1}; Float eta = 1.5f;
    int opTex = -1;                   // uber: "opacity" as an ImageTexture (materials/uber.cpp:53)
    Float glassVRough = 0;            // glass: vroughness (uroughness travels in sigma); both 0: smooth
};
// ImageTexture<RGBSpectrum, Spectrum> with its built MIPMap (textures/imagemap.h, core/mipmap.h).  The pyramid is
// built once by the product's host code (csrc/texture_io.cpp) and travels in the baked scene; lookups are restated here.
enum { WRAP_REPEAT = 0, WRAP_BLACK = 1, WRAP_CLAMP = 2 };
struct MipLevel { int w = 0, h = 0; std::vector<Float> rgb; };
struct Texture {
    std::vector<MipLevel> levels;
    int trilinear = 0; Float maxAniso = 8; int wrap = WRAP_REPEAT;
    Float su = 1, sv = 1, du = 0, dv = 0;
    Float weightLut[128];
};
struct Mesh {
    uint32_t nTris, nVerts;
    bool hasN, hasUV, hasS;
    std::vector<int> idx;
    std::vector<V3> p, n, s;
    std::vector<P2> uv;
};
struct Sphere {
    M44 o2w, w2o;
    Float radius, zMin, zMax, thetaMin, thetaMax, phiMax;
};
struct ShapeRec {
    int kind, material, areaLight, reverseOrientation, swapsHandedness;
    int meshIndex, sphereIndex;   // into Scene::meshes / spheres
    uint32_t nPrims;
    int object = -1;              // >= 0: defined inside ObjectBegin/ObjectEnd (core/api.cpp:1752-1774)
};
struct Light {
    int type;
    V3 pos;      // point: pLight (world); distant: wLight (normalised, world)
    Spec I;      // point: I; distant: L; area: Lemit
    int shape;   // area: shape index
    int twoSided;
    // infinite (lights/infinite.cpp): the radiance map is texture `tex` (texels already multiplied by L * scale, a 1x1 map for a
    // constant light); light <-> world
    int tex = -1;
    M44 l2w, w2l;
};
struct SceneParams {
    int xres, yres;
    Float crop[4];            // x0 x1 y0 y1
    Float filterRadius[2]; int filterType;
    Float filmScale, maxSampleLuminance;
    Float fov, lensRadius, focalDistance, screenWindow[4], shutterOpen, shutterClose;
    M44 camToWorld, worldToCam;
    int spp, samplePixelCenter;
    int maxDepth; Float rrThreshold; int lightStrategy;
    int maxNodePrims, isectCost, travCost;
};
// One entry per primitive of an aggregate in creation order (what the reference's
// renderOptions->primitives / renderOptions->instances[name] hold, core/api.cpp:1638-1649):
// shape >= 0: GeometricPrimitive over that shape's primitive `local`;
// shape == -1: TransformedPrimitive, `local` indexes Scene::instances (core/api.cpp:1817-1819)
struct PrimRef { int shape; int local; };
struct Instance { int object; M44 i2w, w2i; };

struct Scene {
    SceneParams prm;
    std::vector<Material> materials;
    std::vector<ShapeRec> shapes;
    std::vector<Mesh> meshes;
    std::vector<Sphere> spheres;
    std::vector<Light> lights;
    std::vector<Texture> textures;
    std::vector<PrimRef> prims;                          // top level
    std::vector<std::vector<PrimRef>> objectPrims;       // per object definition
    std::vector<Instance> instances;
};

struct Reader {
    FILE *f; bool ok;
    explicit Reader(const char *path) : f(fopen(path, "rb")), ok(f != nullptr) {}
    ~Reader() { if (f) fclose(f); }
    void raw(void *dst, size_t n) { if (ok && fread(dst, 1, n, f) != n) ok = false; }
    int32_t i32() { int32_t v = 0; raw(&v, 4); return v; }
    uint32_t u32() { uint32_t v = 0; raw(&v, 4); return v; }
    float f32() { float v = 0; raw(&v, 4); return v; }
};

inline bool LoadScene(const char *path, Scene *sc, std::string *err) {
    Reader r(path);
    if (!r.ok) { *err = std::string("cannot open ") + path; return false; }
    char magic[8]; r.raw(magic, 8);
    if (!r.ok || memcmp(magic, "HPRTSCN1", 8) != 0) { *err = "bad magic"; return false; }
    uint32_t version = r.u32();
    if (version < 1 || version > 6) { *err = "bad version"; return false; }
    SceneParams &p = sc->prm;
    p.xres = r.i32(); p.yres = r.i32();
    for (int i = 0; i < 4; ++i) p.crop[i] = r.f32();
    p.filterRadius[0] = r.f32(); p.filterRadius[1] = r.f32(); p.filterType = r.i32();
    p.filmScale = r.f32(); p.maxSampleLuminance = r.f32();
    p.fov = r.f32(); p.lensRadius = r.f32(); p.focalDistance = r.f32();
    for (int i = 0; i < 4; ++i) p.screenWindow[i] = r.f32();
    p.shutterOpen = r.f32(); p.shutterClose = r.f32();
    r.raw(p.camToWorld.m, 64); r.raw(p.worldToCam.m, 64);
    p.spp = r.i32(); p.samplePixelCenter = r.i32();
    p.maxDepth = r.i32(); p.rrThreshold = r.f32(); p.lightStrategy = r.i32();
    p.maxNodePrims = r.i32(); p.isectCost = r.i32(); p.travCost = r.i32();
    uint32_t nMat = r.u32(), nShapes = r.u32(), nLights = r.u32();
    if (!r.ok) { *err = "truncated header"; return false; }
    sc->materials.resize(nMat);
    for (auto &m : sc->materials) {
        m.type = r.i32(); r.raw(m.Kd, 12); m.sigma = r.f32(); r.raw(m.Ks, 12);
        m.roughness = r.f32(); m.remap = r.i32();
    }
    sc->shapes.resize(nShapes);
    for (uint32_t si = 0; si < nShapes; ++si) {
        ShapeRec &s = sc->shapes[si];
        s.kind = r.i32(); s.material = r.i32(); s.areaLight = r.i32();
        s.reverseOrientation = r.i32(); s.swapsHandedness = r.i32();
        s.meshIndex = s.sphereIndex = -1;
        if (s.kind == SHAPE_MESH) {
            Mesh m;
            m.nTris = r.u32(); m.nVerts = r.u32();
            uint32_t flags = r.u32();
            m.hasN = flags & 1; m.hasUV = flags & 2; m.hasS = flags & 4;
            if (!r.ok || m.nTris > (1u << 28) || m.nVerts > (1u << 28)) { *err = "bad mesh"; return false; }
            m.idx.resize(3 * (size_t)m.nTris); r.raw(m.idx.data(), 12 * (size_t)m.nTris);
            m.p.resize(m.nVerts); r.raw(m.p.data(), 12 * (size_t)m.nVerts);
            if (m.hasN) { m.n.resize(m.nVerts); r.raw(m.n.data(), 12 * (size_t)m.nVerts); }
            if (m.hasUV) { m.uv.resize(m.nVerts); r.raw(m.uv.data(), 8 * (size_t)m.nVerts); }
            if (m.hasS) { m.s.resize(m.nVerts); r.raw(m.s.data(), 12 * (size_t)m.nVerts); }
            s.meshIndex = (int)sc->meshes.size();
            s.nPrims = m.nTris;
            sc->meshes.push_back(std::move(m));
        } else if (s.kind == SHAPE_SPHERE) {
            Sphere sp;
            r.raw(sp.o2w.m, 64); r.raw(sp.w2o.m, 64);
            sp.radius = r.f32(); sp.zMin = r.f32(); sp.zMax = r.f32();
            sp.thetaMin = r.f32(); sp.thetaMax = r.f32(); sp.phiMax = r.f32();
            s.sphereIndex = (int)sc->spheres.size();
            s.nPrims = 1;
            sc->spheres.push_back(sp);
        } else { *err = "bad shape kind"; return false; }
    }
    sc->lights.resize(nLights);
    for (auto &l : sc->lights) {
        l.type = r.i32();
        float a[3]; r.raw(a, 12); l.pos = V3(a[0], a[1], a[2]);
        float b[3]; r.raw(b, 12); l.I = Spec(b[0], b[1], b[2]);
        l.shape = r.i32(); l.twoSided = r.i32();
    }
    // instancing section (version 2): objects, instances, creation order of the top level
    auto addShapePrims = [&](std::vector<PrimRef> &dst, uint32_t si) {
        for (uint32_t k = 0; k < sc->shapes[si].nPrims; ++k) dst.push_back(PrimRef{(int)si, (int)k});
    };
    if (version >= 2) {
        uint32_t nObjects = r.u32();
        if (!r.ok || nObjects > (1u << 24)) { *err = "bad object count"; return false; }
        sc->objectPrims.resize(nObjects);
        for (uint32_t si = 0; si < nShapes; ++si) {
            int o = r.i32();
            if (o < -1 || o >= (int)nObjects) { *err = "bad object index"; return false; }
            sc->shapes[si].object = o;
            if (o >= 0) addShapePrims(sc->objectPrims[o], si);
        }
        uint32_t nInst = r.u32();
        if (!r.ok || nInst > (1u << 28)) { *err = "bad instance count"; return false; }
        sc->instances.resize(nInst);
        for (auto &in : sc->instances) {
            in.object = r.i32(); r.raw(in.i2w.m, 64); r.raw(in.w2i.m, 64);
            if (in.object < 0 || in.object >= (int)nObjects) { *err = "bad instance object"; return false; }
        }
        uint32_t nTop = r.u32();
        if (!r.ok || nTop > (1u << 28)) { *err = "bad top-level list"; return false; }
        for (uint32_t t = 0; t < nTop; ++t) {
            int kind = r.i32(); uint32_t index = r.u32();
            if (kind == 0) { if (index >= nShapes) { *err = "bad top-level item"; return false; } addShapePrims(sc->prims, index); }
            else { if (kind != 1 || index >= nInst) { *err = "bad top-level item"; return false; } sc->prims.push_back(PrimRef{-1, (int)index}); }
        }
    } else {
        for (uint32_t si = 0; si < nShapes; ++si) addShapePrims(sc->prims, si);
    }
    if (version >= 3) {
        for (auto &m : sc->materials) { m.KdTex = r.i32(); m.KsTex = r.i32(); }
        uint32_t nTex = r.u32();
        if (!r.ok || nTex > (1u << 20)) { *err = "bad texture count"; return false; }
        sc->textures.resize(nTex);
        for (auto &t : sc->textures) {
            // version 6: a texture record is led by its form; the oracle reads finished pyramids only (form 0) — a compact file (the
            // source image, rebuilt by the product's MIPMap constructor at load) is expanded by the product first (Model.save)
            if (version >= 6 && r.i32() != 0) { *err = "compact texture record: save the model in expanded form for the oracle"; return false; }
            t.trilinear = r.i32(); t.maxAniso = r.f32(); t.wrap = r.i32(); t.su = r.f32(); t.sv = r.f32(); t.du = r.f32(); t.dv = r.f32();
            r.raw(t.weightLut, sizeof(t.weightLut));
            uint32_t nl = r.u32();
            if (!r.ok || nl == 0 || nl > 32) { *err = "bad texture header"; return false; }
            t.levels.resize(nl);
            for (auto &l : t.levels) {
                l.w = r.i32(); l.h = r.i32();
                if (!r.ok || l.w <= 0 || l.h <= 0 || l.w > 65536 || l.h > 65536) { *err = "bad texture level"; return false; }
                l.rgb.resize(3 * (size_t)l.w * l.h); r.raw(l.rgb.data(), 4 * l.rgb.size());
            }
        }
        for (auto &m : sc->materials)
            if (m.KdTex >= (int)nTex || m.KsTex >= (int)nTex) { *err = "bad material texture index"; return false; }
    }
    if (version >= 4) {      // infinite lights: map + transform
        for (auto &l : sc->lights)
            if (l.type == LIGHT_INFINITE) {
                l.tex = r.i32(); r.raw(l.l2w.m, 64); r.raw(l.w2l.m, 64);
                if (!r.ok || l.tex < 0 || l.tex >= (int)sc->textures.size()) { *err = "bad infinite light"; return false; }
            }
    } else
        for (auto &l : sc->lights) if (l.type == LIGHT_INFINITE) { *err = "infinite light in a container older than version 4"; return false; }
    if (version >= 5) {      // uber materials: Kr, Kt, opacity, eta
        for (auto &m : sc->materials)
            if (m.type == MAT_UBER) { r.raw(m.Kr, 12); r.raw(m.Kt, 12); r.raw(m.opacity, 12); m.eta = r.f32(); }
    } else
        for (auto &m : sc->materials) if (m.type == MAT_UBER) { *err = "uber material in a container older than version 5"; return false; }
    if (version >= 6)        // uber materials: the opacity texture
        for (auto &m : sc->materials)
            if (m.type == MAT_UBER) {
                m.opTex = r.i32();
                if (!r.ok || m.opTex < -1 || m.opTex >= (int)sc->textures.size()) { *err = "bad opacity texture index"; return false; }
            }
    if (version >= 6)        // glass materials: vroughness
        for (auto &m : sc->materials)
            if (m.type == MAT_GLASS) m.glassVRough = r.f32();
    if (!r.ok) { *err = "truncated file"; return false; }
    return true;
}

}  // namespace orc
