// hprt host side — baked scene container ("HPRTSCN1") writer/reader and a small
// PLY mesh reader (the subset shapes/plymesh.cpp consumes: float x y z [nx ny nz]
// [u v | s t], faces as a uint8/int list of 3 or 4 indices; quads split 0-1-2, 3-0-2
// as at shapes/plymesh.cpp:143-152).
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <fstream>
#include <sstream>
#include "scene_model.h"

namespace hprt {
namespace {

struct Out {
    FILE *f; bool ok = true;
    void raw(const void *p, size_t n) { if (ok && n && fwrite(p, 1, n, f) != n) ok = false; }
    void i32(int32_t v) { raw(&v, 4); }
    void u32(uint32_t v) { raw(&v, 4); }
    void f32(float v) { raw(&v, 4); }
};
struct In {
    FILE *f; bool ok = true;
    void raw(void *p, size_t n) { if (ok && n && fread(p, 1, n, f) != n) ok = false; }
    int32_t i32() { int32_t v = 0; raw(&v, 4); return v; }
    uint32_t u32() { uint32_t v = 0; raw(&v, 4); return v; }
    float f32() { float v = 0; raw(&v, 4); return v; }
};

}  // namespace

bool SaveBakedScene(const SceneModel &sc, const std::string &path, std::string *err, bool compactTextures) {
    FILE *fp = fopen(path.c_str(), "wb");
    if (!fp) { *err = "cannot create " + path; return false; }
    Out o{fp};
    const RenderOptions &p = sc.opt;
    // version 1: every shape is a top-level primitive source.  Version 2 appends the instancing
    // section (objects, instances, top-level creation order) and is written only when needed.
    const bool instancing = sc.nObjects > 0 || !sc.instances.empty();
    // version 3 = version 2's layout (instancing section always present) followed by the image textures
    bool infinite = false, uber = false;
    for (const LightDesc &l : sc.lights) if (l.type == kInfiniteLight) infinite = true;
    for (const MaterialDesc &m : sc.materials) if (m.type == kUber) uber = true;
    // (versions 4 and 5 append to version 3's layout, so its sections are written even when they are empty)
    const bool textured = !sc.textures.empty() || infinite || uber;
    // version 4 = version 3's layout followed by the map index and the transforms of the infinite lights (in light order);
    // version 5 = version 4's followed by Kr, Kt, opacity, eta of the uber materials (in material order)
    // version 6 = version 5's layout with (a) every texture record led by its FORM — 0: the finished levels as before; 1 / 2: the
    // image the levels are built from (8-bit or float texels as ReadImage returned them + ImageTexture::GetTexture's conversion
    // parameters), rebuilt by the product's own MIPMap constructor at load (texture_io.cpp) — and (b) the opacity texture of every
    // uber material after the uber block.  Written when a compact file is asked for or an uber material has a textured opacity.
    bool opacityTextured = false;
    for (const MaterialDesc &m : sc.materials) if (m.type == kUber && m.opacityTex >= 0) opacityTextured = true;
    bool roughGlass = false;      // (c) version 6 also stores every glass material's vroughness (uroughness sits in sigma): rough dielectrics
    for (const MaterialDesc &m : sc.materials) if (m.type == kGlass && (m.sigma != 0.f || m.Kr[0] != 0.f)) roughGlass = true;
    const bool v6 = opacityTextured || roughGlass || (compactTextures && !sc.textures.empty());
    o.raw("HPRTSCN1", 8); o.u32(v6 ? 6u : uber ? 5u : infinite ? 4u : textured ? 3u : instancing ? 2u : 1u);
    o.i32(p.xres); o.i32(p.yres);
    o.raw(p.crop, 16);
    o.raw(p.filterRadius, 8); o.i32(p.filterType);
    o.f32(p.filmScale); o.f32(p.maxSampleLuminance);
    o.f32(p.fov); o.f32(p.lensRadius); o.f32(p.focalDistance);
    o.raw(p.screenWindow, 16);
    o.f32(p.shutterOpen); o.f32(p.shutterClose);
    o.raw(p.cameraToWorld.m, 64); o.raw(p.worldToCamera.m, 64);
    o.i32(p.spp); o.i32(p.samplePixelCenter);
    o.i32(p.maxDepth); o.f32(p.rrThreshold); o.i32(p.lightStrategy);
    o.i32(p.maxNodePrims); o.i32(p.isectCost); o.i32(p.travCost);
    o.u32((uint32_t)sc.materials.size()); o.u32((uint32_t)sc.shapes.size()); o.u32((uint32_t)sc.lights.size());
    for (const MaterialDesc &m : sc.materials) {
        o.i32(m.type); o.raw(m.Kd, 12); o.f32(m.sigma); o.raw(m.Ks, 12); o.f32(m.roughness); o.i32(m.remapRoughness);
    }
    for (const ShapeDesc &s : sc.shapes) {
        o.i32(s.kind); o.i32(s.material); o.i32(s.areaLight); o.i32(s.reverseOrientation); o.i32(s.transformSwapsHandedness);
        if (s.kind == kTriangleMesh) {
            const MeshData &m = s.mesh;
            uint32_t flags = (m.N.empty() ? 0u : 1u) | (m.UV.empty() ? 0u : 2u) | (m.S.empty() ? 0u : 4u);
            o.u32(m.nTris()); o.u32(m.nVerts()); o.u32(flags);
            o.raw(m.indices.data(), 4 * m.indices.size());
            o.raw(m.P.data(), 4 * m.P.size());
            o.raw(m.N.data(), 4 * m.N.size());
            o.raw(m.UV.data(), 4 * m.UV.size());
            o.raw(m.S.data(), 4 * m.S.size());
        } else {
            const SphereData &sp = s.sphere;
            o.raw(sp.objectToWorld.m, 64); o.raw(sp.worldToObject.m, 64);
            o.f32(sp.radius); o.f32(sp.zMin); o.f32(sp.zMax); o.f32(sp.thetaMin); o.f32(sp.thetaMax); o.f32(sp.phiMax);
        }
    }
    for (const LightDesc &l : sc.lights) { o.i32(l.type); o.raw(l.pos, 12); o.raw(l.I, 12); o.i32(l.shape); o.i32(l.twoSided); }
    if (instancing || textured || v6) {
        o.u32(sc.nObjects);
        for (const ShapeDesc &s : sc.shapes) o.i32(s.object);
        o.u32((uint32_t)sc.instances.size());
        for (const InstanceDesc &in : sc.instances) { o.i32(in.object); o.raw(in.instanceToWorld.m, 64); o.raw(in.worldToInstance.m, 64); }
        o.u32((uint32_t)sc.top.size());
        for (const TopItem &t : sc.top) { o.i32(t.kind); o.u32(t.index); }
    }
    if (textured || v6) {
        for (const MaterialDesc &m : sc.materials) { o.i32(m.KdTex); o.i32(m.KsTex); }
        o.u32((uint32_t)sc.textures.size());
        for (const TextureDesc &t : sc.textures) {
            const bool asSource = v6 && compactTextures && (!t.src8.empty() || !t.srcF.empty());
            if (v6) o.i32(asSource ? (t.src8.empty() ? 2 : 1) : 0);
            o.i32(t.trilinear); o.f32(t.maxAniso); o.i32(t.wrap); o.f32(t.su); o.f32(t.sv); o.f32(t.du); o.f32(t.dv);
            if (asSource) {
                o.i32(t.srcW); o.i32(t.srcH); o.f32(t.srcScale); o.i32(t.srcGamma); o.i32(t.srcFlipY);
                if (!t.src8.empty()) o.raw(t.src8.data(), t.src8.size()); else o.raw(t.srcF.data(), 4 * t.srcF.size());
                continue;
            }
            o.raw(t.weightLut, sizeof(t.weightLut));
            o.u32((uint32_t)t.levels.size());
            for (const MipLevel &l : t.levels) { o.i32(l.w); o.i32(l.h); o.raw(l.rgb.data(), 4 * l.rgb.size()); }
        }
    }
    if (infinite || uber || v6)
        for (const LightDesc &l : sc.lights)
            if (l.type == kInfiniteLight) { o.i32(l.texture); o.raw(&l.lightToWorld, 64); o.raw(&l.worldToLight, 64); }
    if (uber || v6)
        for (const MaterialDesc &m : sc.materials)
            if (m.type == kUber) { o.raw(m.Kr, 12); o.raw(m.Kt, 12); o.raw(m.opacity, 12); o.f32(m.eta); }
    if (v6) {
        for (const MaterialDesc &m : sc.materials)
            if (m.type == kUber) o.i32(m.opacityTex);
        for (const MaterialDesc &m : sc.materials)
            if (m.type == kGlass) o.f32(m.Kr[0]);
    }
    bool ok = o.ok;
    if (fclose(fp) != 0) ok = false;
    if (!ok) *err = "write error on " + path;
    return ok;
}

bool LoadBakedScene(const std::string &path, SceneModel *sc, std::string *err) {
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp) { *err = "cannot open " + path; return false; }
    In in{fp};
    auto fail = [&](const char *m) { *err = path + ": " + m; fclose(fp); return false; };
    // sizes read from the file are checked against what the file can still hold BEFORE anything is allocated for them
    fseek(fp, 0, SEEK_END); const long fileSize = ftell(fp); fseek(fp, 0, SEEK_SET);
    auto fits = [&](uint64_t bytes) { const long at = ftell(fp); return at >= 0 && fileSize >= at && bytes <= (uint64_t)(fileSize - at); };
    char magic[8]; in.raw(magic, 8);
    if (!in.ok || memcmp(magic, "HPRTSCN1", 8) != 0) return fail("not a baked hprt scene");
    const uint32_t version = in.u32();
    if (version < 1 || version > 6) return fail("unsupported version");
    RenderOptions &p = sc->opt;
    p.xres = in.i32(); p.yres = in.i32();
    in.raw(p.crop, 16);
    in.raw(p.filterRadius, 8); p.filterType = in.i32();
    p.filmScale = in.f32(); p.maxSampleLuminance = in.f32();
    p.fov = in.f32(); p.lensRadius = in.f32(); p.focalDistance = in.f32();
    in.raw(p.screenWindow, 16);
    p.shutterOpen = in.f32(); p.shutterClose = in.f32();
    in.raw(p.cameraToWorld.m, 64); in.raw(p.worldToCamera.m, 64);
    p.spp = in.i32(); p.samplePixelCenter = in.i32();
    p.maxDepth = in.i32(); p.rrThreshold = in.f32(); p.lightStrategy = in.i32();
    p.maxNodePrims = in.i32(); p.isectCost = in.i32(); p.travCost = in.i32();
    uint32_t nMat = in.u32(), nShapes = in.u32(), nLights = in.u32();
    if (!in.ok || nMat > (1u << 24) || nShapes > (1u << 24) || nLights > (1u << 24)) return fail("truncated or corrupt header");
    if (!fits(32ull * nMat + 20ull * nShapes + 36ull * nLights)) return fail("truncated or corrupt header (counts exceed the file)");
    sc->materials.resize(nMat);
    for (MaterialDesc &m : sc->materials) {
        memset(&m, 0, sizeof(m));
        m.type = in.i32(); in.raw(m.Kd, 12); m.sigma = in.f32(); in.raw(m.Ks, 12); m.roughness = in.f32(); m.remapRoughness = in.i32();
        m.opacity[0] = m.opacity[1] = m.opacity[2] = 1.f; m.eta = 1.5f; m.opacityTex = -1;
        if (m.type < 0 || m.type > kUber || (m.type == kUber && version < 5)) return fail("material type out of range");
    }
    sc->shapes.resize(nShapes);
    for (ShapeDesc &s : sc->shapes) {
        s.kind = in.i32(); s.material = in.i32(); s.areaLight = in.i32(); s.reverseOrientation = in.i32(); s.transformSwapsHandedness = in.i32();
        if (s.material < 0 || (uint32_t)s.material >= nMat) return fail("material index out of range");
        if (s.kind == kTriangleMesh) {
            uint32_t nt = in.u32(), nv = in.u32(), flags = in.u32();
            if (!in.ok || nt > (1u << 28) || nv > (1u << 28)) return fail("corrupt mesh header");
            if (!fits(12ull * nt + 12ull * nv + ((flags & 1) ? 12ull * nv : 0) + ((flags & 2) ? 8ull * nv : 0) + ((flags & 4) ? 12ull * nv : 0))) return fail("truncated mesh");
            MeshData &m = s.mesh;
            m.indices.resize(3 * (size_t)nt); in.raw(m.indices.data(), 12 * (size_t)nt);
            m.P.resize(3 * (size_t)nv); in.raw(m.P.data(), 12 * (size_t)nv);
            if (flags & 1) { m.N.resize(3 * (size_t)nv); in.raw(m.N.data(), 12 * (size_t)nv); }
            if (flags & 2) { m.UV.resize(2 * (size_t)nv); in.raw(m.UV.data(), 8 * (size_t)nv); }
            if (flags & 4) { m.S.resize(3 * (size_t)nv); in.raw(m.S.data(), 12 * (size_t)nv); }
            for (int32_t i : m.indices) if (i < 0 || (uint32_t)i >= nv) return fail("vertex index out of range");
        } else if (s.kind == kSphere) {
            SphereData &sp = s.sphere;
            in.raw(sp.objectToWorld.m, 64); in.raw(sp.worldToObject.m, 64);
            sp.radius = in.f32(); sp.zMin = in.f32(); sp.zMax = in.f32(); sp.thetaMin = in.f32(); sp.thetaMax = in.f32(); sp.phiMax = in.f32();
        } else return fail("unknown shape kind");
    }
    sc->lights.resize(nLights);
    for (LightDesc &l : sc->lights) {
        memset(&l, 0, sizeof(l));
        l.type = in.i32(); in.raw(l.pos, 12); in.raw(l.I, 12); l.shape = in.i32(); l.twoSided = in.i32();
        l.texture = -1;
        if (l.type < 0 || l.type > kInfiniteLight || (l.type == kInfiniteLight && version < 4)) return fail("light type out of range");
        if (l.type == kDiffuseAreaLight && (l.shape < 0 || (uint32_t)l.shape >= nShapes)) return fail("area light shape out of range");
    }
    sc->nObjects = 0; sc->instances.clear(); sc->top.clear();
    if (version >= 2) {
        sc->nObjects = in.u32();
        for (ShapeDesc &s : sc->shapes) { s.object = in.i32(); if (s.object < -1 || s.object >= (int32_t)sc->nObjects) return fail("object index out of range"); }
        const uint32_t nInst = in.u32();
        if (!in.ok || nInst > (1u << 28) || !fits(132ull * nInst)) return fail("corrupt instance count");
        sc->instances.resize(nInst);
        for (InstanceDesc &i : sc->instances) {
            i.object = in.i32(); in.raw(i.instanceToWorld.m, 64); in.raw(i.worldToInstance.m, 64);
            if (i.object < 0 || i.object >= (int32_t)sc->nObjects) return fail("instance object out of range");
        }
        const uint32_t nTop = in.u32();
        if (!in.ok || nTop > (1u << 28) || !fits(8ull * nTop)) return fail("corrupt top-level list");
        sc->top.resize(nTop);
        for (TopItem &t : sc->top) {
            t.kind = in.i32(); t.index = in.u32();
            if (t.kind == 0 ? (t.index >= nShapes || sc->shapes[t.index].object >= 0) : (t.kind != 1 || t.index >= nInst)) return fail("corrupt top-level item");
        }
    } else {
        for (uint32_t i = 0; i < nShapes; ++i) { sc->shapes[i].object = -1; sc->top.push_back(TopItem{0, i}); }
    }
    sc->textures.clear();
    for (MaterialDesc &m : sc->materials) m.KdTex = m.KsTex = -1;
    if (version >= 3) {
        for (MaterialDesc &m : sc->materials) { m.KdTex = in.i32(); m.KsTex = in.i32(); }
        const uint32_t nTex = in.u32();
        if (!in.ok || nTex > (1u << 20)) return fail("corrupt texture count");
        sc->textures.resize(nTex);
        for (TextureDesc &t : sc->textures) {
            const int32_t form = version >= 6 ? in.i32() : 0;
            t.trilinear = in.i32(); t.maxAniso = in.f32(); t.wrap = in.i32(); t.su = in.f32(); t.sv = in.f32(); t.du = in.f32(); t.dv = in.f32();
            if (!in.ok || form < 0 || form > 2 || t.wrap < 0 || t.wrap > 2) return fail("corrupt texture header");
            if (form != 0) {      // the source image: the levels are rebuilt here, by the constructor that built them at parse time
                t.srcW = in.i32(); t.srcH = in.i32(); t.srcScale = in.f32(); t.srcGamma = in.i32(); t.srcFlipY = in.i32();
                if (!in.ok || t.srcW <= 0 || t.srcH <= 0 || t.srcW > 65536 || t.srcH > 65536) return fail("corrupt texture source");
                const uint64_t nv = 3ull * (uint64_t)t.srcW * (uint64_t)t.srcH;
                if (!fits(form == 1 ? nv : 4 * nv)) return fail("truncated texture source");
                if (form == 1) { t.src8.resize(nv); in.raw(t.src8.data(), nv); } else { t.srcF.resize(nv); in.raw(t.srcF.data(), 4 * nv); }
                if (!in.ok) return fail("truncated texture source");
                RebuildFromSource(&t);
                continue;
            }
            in.raw(t.weightLut, sizeof(t.weightLut));
            const uint32_t nl = in.u32();
            if (!in.ok || nl == 0 || nl > 32 || t.wrap < 0 || t.wrap > 2) return fail("corrupt texture header");
            t.levels.resize(nl);
            for (MipLevel &l : t.levels) {
                l.w = in.i32(); l.h = in.i32();
                if (!in.ok || l.w <= 0 || l.h <= 0 || l.w > 65536 || l.h > 65536 || !fits(12ull * (uint64_t)l.w * (uint64_t)l.h)) return fail("corrupt texture level");
                l.rgb.resize(3 * (size_t)l.w * l.h); in.raw(l.rgb.data(), 4 * l.rgb.size());
            }
        }
        for (const MaterialDesc &m : sc->materials)
            if (m.KdTex >= (int32_t)nTex || m.KsTex >= (int32_t)nTex || m.KdTex < -1 || m.KsTex < -1) return fail("material texture index out of range");
    }
    if (version >= 4)
        for (LightDesc &l : sc->lights)
            if (l.type == kInfiniteLight) {
                l.texture = in.i32(); in.raw(&l.lightToWorld, 64); in.raw(&l.worldToLight, 64);
                if (!in.ok || l.texture < 0 || (size_t)l.texture >= sc->textures.size()) return fail("infinite light map out of range");
            }
    if (version >= 5)
        for (MaterialDesc &m : sc->materials)
            if (m.type == kUber) { in.raw(m.Kr, 12); in.raw(m.Kt, 12); in.raw(m.opacity, 12); m.eta = in.f32(); }
    if (version >= 6)
        for (MaterialDesc &m : sc->materials)
            if (m.type == kUber) {
                m.opacityTex = in.i32();
                if (!in.ok || m.opacityTex < -1 || m.opacityTex >= (int32_t)sc->textures.size()) return fail("opacity texture index out of range");
            }
    if (version >= 6)
        for (MaterialDesc &m : sc->materials)
            if (m.type == kGlass) m.Kr[0] = in.f32();
    if (!in.ok) return fail("truncated file");
    fclose(fp);
    return true;
}

bool ReadPlyMesh(const std::string &path, std::vector<int> *idx, std::vector<float> *P, std::vector<float> *N,
                 std::vector<float> *UV, std::string *err) {
    std::ifstream in(path, std::ios::binary);
    if (!in) { *err = "Couldn't open PLY file \"" + path + "\""; return false; }
    std::string line;
    std::getline(in, line);
    if (line.substr(0, 3) != "ply") { *err = path + ": not a PLY file"; return false; }
    bool binary = false, bigEndian = false;
    struct Prop { std::string name, type, countType; bool list = false; };
    struct Elem { std::string name; long count = 0; std::vector<Prop> props; };
    std::vector<Elem> elems;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ls(line);
        std::string w; ls >> w;
        if (w == "format") { std::string f; ls >> f; if (f == "binary_little_endian") binary = true; else if (f == "binary_big_endian") { binary = true; bigEndian = true; } else if (f != "ascii") { *err = path + ": unsupported PLY format " + f; return false; } }
        else if (w == "element") { Elem e; ls >> e.name >> e.count; elems.push_back(e); }
        else if (w == "property" && !elems.empty()) {
            Prop p; std::string t; ls >> t;
            if (t == "list") { p.list = true; ls >> p.countType >> p.type >> p.name; } else { p.type = t; ls >> p.name; }
            elems.back().props.push_back(p);
        } else if (w == "end_header") break;
    }
    auto tsize = [](const std::string &t) -> int {
        if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
        if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
        if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
        if (t == "double" || t == "float64") return 8;
        return 0;
    };
    auto readScalar = [&](const std::string &t, double *out) -> bool {
        if (!binary) { return (bool)(in >> *out); }
        unsigned char b[8]; int n = tsize(t);
        if (n == 0 || !in.read((char *)b, n)) return false;
        if (bigEndian) std::reverse(b, b + n);
        if (t == "float" || t == "float32") { float f; memcpy(&f, b, 4); *out = f; }
        else if (t == "double" || t == "float64") { double d; memcpy(&d, b, 8); *out = d; }
        else if (t == "uchar" || t == "uint8") *out = b[0];
        else if (t == "char" || t == "int8") *out = (signed char)b[0];
        else if (t == "ushort" || t == "uint16") { uint16_t v; memcpy(&v, b, 2); *out = v; }
        else if (t == "short" || t == "int16") { int16_t v; memcpy(&v, b, 2); *out = v; }
        else if (t == "uint" || t == "uint32") { uint32_t v; memcpy(&v, b, 4); *out = v; }
        else { int32_t v; memcpy(&v, b, 4); *out = v; }
        return true;
    };
    const std::streampos dataPos = in.tellg();
    in.seekg(0, std::ios::end); const long long fileSize = (long long)in.tellg(); in.seekg(dataPos);
    for (const Elem &e : elems) {
        // an element cannot have more entries than the file has bytes: refuse before allocating for it
        if (e.count < 0 || e.count > (1l << 28) || (long long)e.count > fileSize) { *err = path + ": PLY element count out of range"; return false; }
        if (e.name == "vertex") {
            int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1, iu = -1, iv = -1;
            for (size_t k = 0; k < e.props.size(); ++k) {
                const std::string &n = e.props[k].name;
                if (n == "x") ix = (int)k; else if (n == "y") iy = (int)k; else if (n == "z") iz = (int)k;
                else if (n == "nx") inx = (int)k; else if (n == "ny") iny = (int)k; else if (n == "nz") inz = (int)k;
                else if (n == "u" || n == "s" || n == "texture_u" || n == "texture_s") iu = (int)k;
                else if (n == "v" || n == "t" || n == "texture_v" || n == "texture_t") iv = (int)k;
            }
            if (ix < 0 || iy < 0 || iz < 0) { *err = path + ": Vertex coordinate property not found!"; return false; }
            bool hasN = inx >= 0 && iny >= 0 && inz >= 0, hasUV = iu >= 0 && iv >= 0;
            P->resize(3 * (size_t)e.count); if (hasN) N->resize(3 * (size_t)e.count); if (hasUV) UV->resize(2 * (size_t)e.count);
            std::vector<double> vals(e.props.size());
            for (long i = 0; i < e.count; ++i) {
                for (size_t k = 0; k < e.props.size(); ++k) if (!readScalar(e.props[k].type, &vals[k])) { *err = path + ": unable to read the contents of PLY file"; return false; }
                (*P)[3 * i] = (float)vals[ix]; (*P)[3 * i + 1] = (float)vals[iy]; (*P)[3 * i + 2] = (float)vals[iz];
                if (hasN) { (*N)[3 * i] = (float)vals[inx]; (*N)[3 * i + 1] = (float)vals[iny]; (*N)[3 * i + 2] = (float)vals[inz]; }
                if (hasUV) { (*UV)[2 * i] = (float)vals[iu]; (*UV)[2 * i + 1] = (float)vals[iv]; }
            }
        } else {
            for (long i = 0; i < e.count; ++i)
                for (const Prop &p : e.props) {
                    if (p.list) {
                        double c; if (!readScalar(p.countType, &c)) { *err = path + ": truncated face list"; return false; }
                        if (!(c >= 0 && c <= 255)) { *err = path + ": PLY face with an impossible vertex count"; return false; }
                        int n = (int)c; std::vector<int> face(n);
                        for (int k = 0; k < n; ++k) { double v; if (!readScalar(p.type, &v)) { *err = path + ": truncated face list"; return false; } face[k] = (int)v; }
                        if (e.name == "face" && (p.name == "vertex_indices" || p.name == "vertex_index")) {
                            if (n == 3 || n == 4) {
                                for (int k = 0; k < 3; ++k) idx->push_back(face[k]);
                                if (n == 4) { idx->push_back(face[3]); idx->push_back(face[0]); idx->push_back(face[2]); }
                            }
                        }
                    } else { double v; if (!readScalar(p.type, &v)) { *err = path + ": truncated element"; return false; } }
                }
        }
    }
    if (P->empty() || idx->empty()) { *err = path + ": PLY file is invalid! No face/vertex elements found!"; return false; }
    size_t nv = P->size() / 3;
    for (int i : *idx) if (i < 0 || (size_t)i >= nv) { *err = path + ": plymesh: Vertex reference out of bounds"; return false; }
    return true;
}

}  // namespace hprt
