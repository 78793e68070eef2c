// hprt host side — scene-description front-end for the directive subset the bundled
// scenes use.  Mirrors the *behaviour* of the pbrt-v3 fork's
// core/parser.cpp (tokens, typed parameter lists) and core/api.cpp (CTM, graphics
// state, named coordinate systems, shape/light/material creation) and produces a
// SceneModel: world-space shapes in creation order, exactly what MakeScene() hands
// to the accelerator (core/api.cpp:1883-1892).
#include <algorithm>
#include <cstdio>
#include <initializer_list>
#include <cstdlib>
#include <cstring>
#include <strings.h>
#include <fstream>
#include <map>
#include <sstream>
#include "host_transform.h"
#include "scene_model.h"

namespace hprt {

bool LoopSubdivide(int nLevels, const std::vector<int> &indices, const std::vector<float> &P,
                   std::vector<int> *outIdx, std::vector<float> *outP, std::vector<float> *outN, std::string *err);
bool ReadPlyMesh(const std::string &path, std::vector<int> *idx, std::vector<float> *P, std::vector<float> *N,
                 std::vector<float> *UV, std::string *err);

namespace {

struct Param {
    std::string type, name;
    std::vector<float> nums;
    std::vector<std::string> strs;
    std::vector<bool> bools;
    mutable bool used = false;
};
struct ParamList {
    std::vector<Param> items;
    const Param *find(const std::string &name, const char *t1, const char *t2 = nullptr, const char *t3 = nullptr) const {
        for (const Param &p : items)
            if (p.name == name && (p.type == t1 || (t2 && p.type == t2) || (t3 && p.type == t3))) { p.used = true; return &p; }
        return nullptr;
    }
    float oneFloat(const std::string &n, float d) const { const Param *p = find(n, "float"); return (p && p->nums.size() == 1) ? p->nums[0] : d; }
    int oneInt(const std::string &n, int d) const { const Param *p = find(n, "integer"); return (p && p->nums.size() == 1) ? (int)p->nums[0] : d; }
    bool oneBool(const std::string &n, bool d) const { const Param *p = find(n, "bool"); return (p && p->bools.size() == 1) ? p->bools[0] : d; }
    std::string oneString(const std::string &n, const std::string &d) const { const Param *p = find(n, "string"); return (p && p->strs.size() == 1) ? p->strs[0] : d; }
    const std::vector<float> *floats(const std::string &n) const { const Param *p = find(n, "float"); return p ? &p->nums : nullptr; }
    const std::vector<float> *ints(const std::string &n) const { const Param *p = find(n, "integer"); return p ? &p->nums : nullptr; }
    const std::vector<float> *points(const std::string &n) const { const Param *p = find(n, "point", "point3"); return p ? &p->nums : nullptr; }
    const std::vector<float> *vectors(const std::string &n) const { const Param *p = find(n, "vector", "vector3"); return p ? &p->nums : nullptr; }
    const std::vector<float> *normals(const std::string &n) const { const Param *p = find(n, "normal", "normal3"); return p ? &p->nums : nullptr; }
    const std::vector<float> *point2s(const std::string &n) const { const Param *p = find(n, "point2", "vector2"); return p ? &p->nums : nullptr; }
    bool rgb3(const std::string &n, float out[3]) const {
        const Param *p = find(n, "color", "rgb");
        if (p && p->nums.size() == 3) { out[0] = p->nums[0]; out[1] = p->nums[1]; out[2] = p->nums[2]; return true; }
        return false;
    }
    std::string texture(const std::string &n) const { const Param *p = find(n, "texture"); return (p && p->strs.size() == 1) ? p->strs[0] : ""; }
};

struct Tokenizer {
    std::string text; size_t pos = 0; std::string file; int line = 1;
    bool next(std::string *tok) {
        while (pos < text.size()) {
            char c = text[pos];
            if (c == '\n') { ++line; ++pos; }
            else if (c == ' ' || c == '\t' || c == '\r') ++pos;
            else if (c == '#') { while (pos < text.size() && text[pos] != '\n') ++pos; }
            else break;
        }
        if (pos >= text.size()) return false;
        char c = text[pos];
        if (c == '"') {
            size_t e = pos + 1;
            while (e < text.size() && text[e] != '"') { if (text[e] == '\n') ++line; ++e; }
            *tok = text.substr(pos, e - pos + 1);
            pos = e + 1;
        } else if (c == '[' || c == ']') { *tok = std::string(1, c); ++pos; }
        else {
            size_t e = pos;
            while (e < text.size() && !isspace((unsigned char)text[e]) && text[e] != '"' && text[e] != '[' && text[e] != ']') ++e;
            *tok = text.substr(pos, e - pos);
            pos = e;
        }
        return true;
    }
};

struct TexConst { bool isFloat; float v[3]; int image = -1; };   // image >= 0: SceneModel::textures index (spectrum imagemap)

struct GraphicsState {
    std::string materialName = "matte";
    ParamList materialParams;
    std::string namedMaterial;
    std::string areaLight; ParamList areaLightParams;
    bool reverseOrientation = false;
    std::map<std::string, TexConst> textures;
};

struct Frontend {
    SceneModel *sc;
    std::string err;
    std::vector<Tokenizer> files;
    std::string pending; bool hasPending = false;
    Xform ctm;
    std::map<std::string, Xform> namedCS;
    std::vector<GraphicsState> gsStack; GraphicsState gs;
    std::vector<Xform> xfStack;
    std::map<std::string, int> imageCache;         // ImageTexture::textures (textures/imagemap.h:119)
    int currentObject = -1;                        // renderOptions->currentInstance
    std::map<std::string, int> objectByName;       // renderOptions->instances
    std::map<std::string, std::pair<std::string, ParamList>> namedMaterials;
    std::map<std::string, int> materialCache;
    bool inWorld = false;
    std::string cameraName = "perspective"; ParamList cameraParams; Xform cameraToWorld;
    std::string filmName = "image"; ParamList filmParams;
    std::string filterName = "box"; ParamList filterParams;
    ParamList samplerParams, accelParams, integratorParams;
    std::string baseDir;
    std::map<std::string, std::string> subst;   // template tokens ("$acc") and path prefixes ("/old/prefix/")

    bool fail(const std::string &m) {
        std::ostringstream o;
        if (!files.empty()) o << files.back().file << ":" << files.back().line << ": ";
        o << m; err = o.str(); return false;
    }
    void warn(const std::string &m) { if (std::find(sc->warnings.begin(), sc->warnings.end(), m) == sc->warnings.end()) sc->warnings.push_back(m); }
    // ParamSet::ReportUnused (core/paramset.cpp:443-459): a parameter nobody looked up.  Here that also covers what the reference
    // WOULD have read but this front-end does not (alpha textures, bump maps, ...): never silently dropped.  `ignored`: names the
    // reference reads and this path has no use for (file names, sample counts of other integrators).
    void reportUnused(const ParamList &pl, const std::string &what, std::initializer_list<const char *> ignored = {}) {
        for (const Param &p : pl.items) {
            if (p.used) continue;
            bool skip = false;
            for (const char *n : ignored) if (p.name == n) skip = true;
            if (!skip) warn("Parameter \"" + p.type + " " + p.name + "\" of " + what + " not used");
        }
    }

    bool nextToken(std::string *t) {
        if (hasPending) { *t = pending; hasPending = false; return true; }
        while (!files.empty()) {
            if (files.back().next(t)) {
                // the fork's scene files are sed templates (scripts/render_simple.sh:23-29)
                if (!t->empty() && (*t)[0] == '$') {
                    auto it = subst.find(*t);
                    *t = it != subst.end() ? it->second : std::string("0");
                }
                return true;
            }
            files.pop_back();
        }
        return false;
    }
    void unget(const std::string &t) { pending = t; hasPending = true; }
    static bool quoted(const std::string &t) { return t.size() >= 2 && t.front() == '"' && t.back() == '"'; }
    static std::string dequote(const std::string &t) { return t.substr(1, t.size() - 2); }

    // parser.cpp:322-368: integers via strtol, everything else via strtof
    static bool parseNumber(const std::string &s, float *out) {
        if (s.empty()) return false;
        bool isInt = true;
        for (char ch : s) if (!(ch >= '0' && ch <= '9')) { isInt = false; break; }
        char *end = nullptr;
        double v;
        if (isInt) v = (double)strtol(s.c_str(), &end, 10);
        else v = strtof(s.c_str(), &end);
        if (end == s.c_str()) return false;
        *out = (float)v;
        return true;
    }
    bool readFloats(int n, float *out) {
        for (int i = 0; i < n; ++i) {
            std::string t;
            if (!nextToken(&t) || !parseNumber(t, &out[i])) return fail("expected a number");
        }
        return true;
    }
    bool readQuoted(std::string *out) {
        std::string t;
        if (!nextToken(&t) || !quoted(t)) return fail("expected a quoted string");
        *out = dequote(t);
        return true;
    }
    // parser.cpp parseParams: "type name" value | [ values ]
    bool readParams(ParamList *pl) {
        pl->items.clear();
        std::string t;
        while (nextToken(&t)) {
            if (!quoted(t)) { unget(t); break; }
            std::string decl = dequote(t);
            std::istringstream ds(decl);
            Param p;
            if (!(ds >> p.type >> p.name)) return fail("bad parameter declaration \"" + decl + "\"");
            std::vector<std::string> vals;
            std::string v;
            if (!nextToken(&v)) return fail("premature EOF in parameter list");
            if (v == "[") {
                while (true) {
                    if (!nextToken(&v)) return fail("premature EOF in parameter list");
                    if (v == "]") break;
                    vals.push_back(v);
                }
            } else vals.push_back(v);
            for (const std::string &s : vals) {
                if (p.type == "string" || p.type == "texture" || p.type == "spectrum") {
                    if (quoted(s)) p.strs.push_back(dequote(s));
                    else { float f; if (!parseNumber(s, &f)) return fail("bad value " + s); p.nums.push_back(f); }
                } else if (p.type == "bool") {
                    std::string b = quoted(s) ? dequote(s) : s;
                    p.bools.push_back(b == "true");
                } else {
                    float f;
                    if (!parseNumber(s, &f)) return fail("expected a number, got " + s);
                    p.nums.push_back(f);
                }
            }
            // spectra given in other forms (core/paramset.cpp:168-215).  "xyz" is a fixed matrix away from RGB (RGBSpectrum::FromXYZ ->
            // XYZToRGB, core/spectrum.h:52-56); blackbody and sampled spectra need the CIE matching tables, which are outside the
            // hot-path scope: the directive then uses its default, and says so.
            if (p.type == "xyz" && p.nums.size() % 3 == 0) {
                for (size_t k = 0; k + 2 < p.nums.size(); k += 3) {
                    const float x = p.nums[k], y = p.nums[k + 1], z = p.nums[k + 2];
                    p.nums[k] = 3.240479f * x - 1.537150f * y - 0.498535f * z;
                    p.nums[k + 1] = -0.969256f * x + 1.875991f * y + 0.041556f * z;
                    p.nums[k + 2] = 0.055648f * x - 0.204043f * y + 1.057311f * z;
                }
                p.type = "rgb";
            } else if (p.type == "blackbody" || (p.type == "spectrum" && p.name != "eta" && p.name != "k")) {
                p.used = true;      // (reported here, not again as unused)
                warn("parameter \"" + p.type + " " + p.name + "\": spectra given as " + (p.type == "blackbody" ? "blackbody temperatures" : "sampled data or files") +
                     " are outside the hot-path scope; the default value is used");
            }
            pl->items.push_back(p);
        }
        return true;
    }

    // ---- materials (core/api.cpp MakeMaterial + TextureParams lookups) -----
    bool spectrumParam(const ParamList &geom, const ParamList &mat, const std::string &name, const float def[3], float out[3], int32_t *image = nullptr) {
        // TextureParams::GetSpectrumTexture: geometry params first, then material params
        if (image) *image = -1;
        for (const ParamList *pl : {&geom, &mat}) {
            std::string tex = pl->texture(name);
            if (!tex.empty()) {
                auto it = gs.textures.find(tex);
                if (it != gs.textures.end() && !it->second.isFloat && it->second.image >= 0 && image) { *image = it->second.image; memcpy(out, def, 12); return true; }
                if (it != gs.textures.end() && !it->second.isFloat && it->second.image < 0) { memcpy(out, it->second.v, 12); return true; }
                warn("texture \"" + tex + "\" for \"" + name + "\" is not a constant spectrum texture; using the default");
                memcpy(out, def, 12); return true;
            }
            if (pl->rgb3(name, out)) return true;
        }
        memcpy(out, def, 12);
        return true;
    }
    float floatParam(const ParamList &geom, const ParamList &mat, const std::string &name, float def) {
        for (const ParamList *pl : {&geom, &mat}) {
            std::string tex = pl->texture(name);
            if (!tex.empty()) {
                auto it = gs.textures.find(tex);
                if (it != gs.textures.end() && it->second.isFloat) return it->second.v[0];
                warn("texture \"" + tex + "\" for \"" + name + "\" is not a constant float texture; using the default");
                return def;
            }
            const Param *p = pl->find(name, "float");
            if (p && p->nums.size() == 1) return p->nums[0];
        }
        return def;
    }
    int materialForShape(const ParamList &geom) {
        std::string name = gs.materialName;
        const ParamList *mp = &gs.materialParams;
        if (!gs.namedMaterial.empty()) {
            auto it = namedMaterials.find(gs.namedMaterial);
            if (it != namedMaterials.end()) { name = it->second.first; mp = &it->second.second; }
        }
        MaterialDesc m;
        memset(&m, 0, sizeof(m));
        m.KdTex = m.KsTex = m.opacityTex = -1;
        if (name == "plastic") {
            const float dk[3] = {0.25f, 0.25f, 0.25f};
            m.type = kPlastic;
            spectrumParam(geom, *mp, "Kd", dk, m.Kd, &m.KdTex);
            spectrumParam(geom, *mp, "Ks", dk, m.Ks, &m.KsTex);
            m.roughness = floatParam(geom, *mp, "roughness", .1f);
            bool remap = true;
            const Param *rp = geom.find("remaproughness", "bool"); if (!rp) rp = mp->find("remaproughness", "bool");
            if (rp && rp->bools.size() == 1) remap = rp->bools[0];
            m.remapRoughness = remap ? 1 : 0;
        } else if (name == "substrate") {   // CreateSubstrateMaterial, materials/substrate.cpp:67-82
            const float dk[3] = {0.5f, 0.5f, 0.5f};
            m.type = kSubstrate;
            spectrumParam(geom, *mp, "Kd", dk, m.Kd, &m.KdTex);
            spectrumParam(geom, *mp, "Ks", dk, m.Ks, &m.KsTex);
            m.roughness = floatParam(geom, *mp, "uroughness", .1f);
            m.sigma = floatParam(geom, *mp, "vroughness", .1f);
            bool remap = true;
            const Param *rp = geom.find("remaproughness", "bool"); if (!rp) rp = mp->find("remaproughness", "bool");
            if (rp && rp->bools.size() == 1) remap = rp->bools[0];
            m.remapRoughness = remap ? 1 : 0;
        } else if (name == "metal" && (mp->find("eta", "rgb") || mp->find("eta", "color") || geom.find("eta", "rgb") || geom.find("eta", "color"))) {
            // CreateMetalMaterial, materials/metal.cpp:115-134 — with eta and k given as RGB; their defaults are the measured
            // copper SPECTRA (metal.cpp:81-113), which an RGB build converts through its CIE tables: outside this scope
            const float one[3] = {1.f, 1.f, 1.f};
            int dummy = -1;
            m.type = kMetal;
            spectrumParam(geom, *mp, "eta", one, m.Kd, &dummy);
            spectrumParam(geom, *mp, "k", one, m.Ks, &dummy);
            if (!(mp->find("k", "rgb") || mp->find("k", "color") || geom.find("k", "rgb") || geom.find("k", "color")))
                warn("metal without an RGB \"k\": the copper default is spectral data outside the scope; k = 1 used");
            const float rough = floatParam(geom, *mp, "roughness", .01f);
            m.roughness = floatParam(geom, *mp, "uroughness", rough);      // GetFloatTextureOrNull: the common roughness when absent
            m.sigma = floatParam(geom, *mp, "vroughness", rough);
            bool remap = true;
            const Param *rp = geom.find("remaproughness", "bool"); if (!rp) rp = mp->find("remaproughness", "bool");
            if (rp && rp->bools.size() == 1) remap = rp->bools[0];
            m.remapRoughness = remap ? 1 : 0;
        } else if (name == "glass") {       // CreateGlassMaterial, materials/glass.cpp:88-105: smooth (FresnelSpecular) or rough (microfacet) dielectric
            const float one[3] = {1.f, 1.f, 1.f};
            m.type = kGlass;
            spectrumParam(geom, *mp, "Kr", one, m.Ks, &m.KsTex);
            spectrumParam(geom, *mp, "Kt", one, m.Kd, &m.KdTex);
            const bool hasEta = mp->find("eta", "float") || geom.find("eta", "float") || !mp->texture("eta").empty() || !geom.texture("eta").empty();
            m.roughness = hasEta ? floatParam(geom, *mp, "eta", 1.5f) : floatParam(geom, *mp, "index", 1.5f);
            // "uroughness" / "vroughness" (both 0: the smooth FresnelSpecular lobe; else MicrofacetReflection + MicrofacetTransmission,
            // materials/glass.cpp:61-93): uroughness travels in sigma, vroughness in Kr[0]
            m.sigma = floatParam(geom, *mp, "uroughness", 0.f);
            m.Kr[0] = floatParam(geom, *mp, "vroughness", 0.f);
            bool remap = true;
            const Param *rp = geom.find("remaproughness", "bool"); if (!rp) rp = mp->find("remaproughness", "bool");
            if (rp && rp->bools.size() == 1) remap = rp->bools[0];
            m.remapRoughness = remap ? 1 : 0;
        } else if (name == "uber") {        // CreateUberMaterial, materials/uber.cpp:110-140 (constant parameters; Kd / Ks may be image textures)
            const float q[3] = {0.25f, 0.25f, 0.25f}, zero[3] = {0.f, 0.f, 0.f}, one[3] = {1.f, 1.f, 1.f};
            m.type = kUber;
            spectrumParam(geom, *mp, "Kd", q, m.Kd, &m.KdTex);
            spectrumParam(geom, *mp, "Ks", q, m.Ks, &m.KsTex);
            spectrumParam(geom, *mp, "Kr", zero, m.Kr);
            spectrumParam(geom, *mp, "Kt", zero, m.Kt);
            spectrumParam(geom, *mp, "opacity", one, m.opacity, &m.opacityTex);      // (an image texture: the leaves of scenes/livingroom, :30)
            const float rough = floatParam(geom, *mp, "roughness", .1f);
            const bool hasU = mp->find("uroughness", "float") || geom.find("uroughness", "float") || !mp->texture("uroughness").empty() || !geom.texture("uroughness").empty();
            const bool hasV = mp->find("vroughness", "float") || geom.find("vroughness", "float") || !mp->texture("vroughness").empty() || !geom.texture("vroughness").empty();
            m.roughness = hasU ? floatParam(geom, *mp, "uroughness", rough) : rough;      // roughu = uroughness or roughness; roughv = vroughness or roughu (:76-85)
            m.sigma = hasV ? floatParam(geom, *mp, "vroughness", m.roughness) : m.roughness;
            const bool hasEta = mp->find("eta", "float") || geom.find("eta", "float") || !mp->texture("eta").empty() || !geom.texture("eta").empty();
            m.eta = hasEta ? floatParam(geom, *mp, "eta", 1.5f) : floatParam(geom, *mp, "index", 1.5f);
            bool remap = true;
            const Param *rp = geom.find("remaproughness", "bool"); if (!rp) rp = mp->find("remaproughness", "bool");
            if (rp && rp->bools.size() == 1) remap = rp->bools[0];
            m.remapRoughness = remap ? 1 : 0;
        } else if (name == "mirror") {      // CreateMirrorMaterial, materials/mirror.cpp:58-64
            const float dk[3] = {0.9f, 0.9f, 0.9f};
            m.type = kMirror;
            spectrumParam(geom, *mp, "Kr", dk, m.Ks, &m.KsTex);
        } else {
            if (name != "matte" && name != "" && name != "none")
                warn("material \"" + name + "\" is outside the hot-path scope; rendered as matte (SURVEY.md §2)");
            const float dk[3] = {0.5f, 0.5f, 0.5f};
            m.type = kMatte;
            spectrumParam(geom, *mp, "Kd", dk, m.Kd, &m.KdTex);
            m.sigma = floatParam(geom, *mp, "sigma", 0.f);      // != 0: OrenNayar (materials/matte.cpp:55-61)
        }
        reportUnused(*mp, "Material \"" + name + "\"", {"type"});
        std::string key((const char *)&m, sizeof(m));
        auto it = materialCache.find(key);
        if (it != materialCache.end()) return it->second;
        int id = (int)sc->materials.size();
        sc->materials.push_back(m);
        materialCache[key] = id;
        return id;
    }

    // MakeAreaLight, core/api.cpp:782-788 + CreateDiffuseAreaLight, lights/diffuse.cpp:113-125
    LightDesc makeAreaLight() {
        if (gs.areaLight != "area" && gs.areaLight != "diffuse") warn("area light \"" + gs.areaLight + "\" unknown; treated as diffuse");
        LightDesc l; memset(&l, 0, sizeof(l));
        l.type = kDiffuseAreaLight;
        float L[3] = {1, 1, 1}, scv[3] = {1, 1, 1};
        gs.areaLightParams.rgb3("L", L);
        gs.areaLightParams.rgb3("scale", scv);
        for (int i = 0; i < 3; ++i) l.I[i] = L[i] * scv[i];
        l.twoSided = gs.areaLightParams.oneBool("twosided", false) ? 1 : 0;
        reportUnused(gs.areaLightParams, "AreaLightSource", {"nsamples", "samples"});
        return l;
    }

    // ---- shapes (core/api.cpp:1561-1651) ------------------------------------
    bool addMeshShape(const ParamList &params, std::vector<int> &idx, std::vector<float> &P, std::vector<float> &N,
                      std::vector<float> &UV, std::vector<float> &S) {
        ShapeDesc sh;
        sh.kind = kTriangleMesh;
        sh.material = materialForShape(params);
        sh.areaLight = -1;
        sh.reverseOrientation = gs.reverseOrientation ? 1 : 0;
        sh.transformSwapsHandedness = ctm.swapsHandedness() ? 1 : 0;
        size_t nv = P.size() / 3;
        for (int i : idx) if (i < 0 || (size_t)i >= nv) return fail("trianglemesh has out-of-bounds vertex index");
        // TriangleMesh ctor: P -> world (points), N -> world (normals), S -> world (vectors), shapes/triangle.cpp:72-88
        sh.mesh.indices.assign(idx.begin(), idx.end());
        sh.mesh.P.resize(P.size());
        for (size_t i = 0; i < nv; ++i) {
            vec3 p = xf_point(ctm.m, vec3(P[3 * i], P[3 * i + 1], P[3 * i + 2]));
            sh.mesh.P[3 * i] = p.x; sh.mesh.P[3 * i + 1] = p.y; sh.mesh.P[3 * i + 2] = p.z;
        }
        if (N.size() == P.size()) {
            sh.mesh.N.resize(N.size());
            for (size_t i = 0; i < nv; ++i) {
                vec3 n = xf_normal(ctm.inv, vec3(N[3 * i], N[3 * i + 1], N[3 * i + 2]));
                sh.mesh.N[3 * i] = n.x; sh.mesh.N[3 * i + 1] = n.y; sh.mesh.N[3 * i + 2] = n.z;
            }
        }
        if (S.size() == P.size()) {
            sh.mesh.S.resize(S.size());
            for (size_t i = 0; i < nv; ++i) {
                vec3 s = xf_vector(ctm.m, vec3(S[3 * i], S[3 * i + 1], S[3 * i + 2]));
                sh.mesh.S[3 * i] = s.x; sh.mesh.S[3 * i + 1] = s.y; sh.mesh.S[3 * i + 2] = s.z;
            }
        }
        if (UV.size() >= 2 * nv) sh.mesh.UV.assign(UV.begin(), UV.begin() + 2 * nv);
        if (!gs.areaLight.empty() && currentObject >= 0)
            warn("Area lights not supported with object instancing (core/api.cpp:1640); the shape is kept without emission");
        else if (!gs.areaLight.empty() && !idx.empty()) {
            // pbrtShape creates one Triangle shape per face and one DiffuseAreaLight per shape, appended to the scene's lights
            // in face order (core/api.cpp:1609-1636): the mesh owns the consecutive lights areaLight .. areaLight + nTris - 1
            LightDesc l = makeAreaLight();
            l.shape = (int)sc->shapes.size();
            sh.areaLight = (int)sc->lights.size();
            for (size_t t = 0; t < idx.size() / 3; ++t) sc->lights.push_back(l);
        }
        pushShape(std::move(sh));
        return true;
    }
    // "Add prims to scene or current instance", core/api.cpp:1638-1650
    void pushShape(ShapeDesc &&sh) {
        sh.object = currentObject;
        if (currentObject < 0) sc->top.push_back(TopItem{0, (uint32_t)sc->shapes.size()});
        sc->shapes.push_back(std::move(sh));
    }
    bool doShape(const std::string &name, const ParamList &params) {
        if (name == "trianglemesh" || name == "loopsubdiv" || name == "plymesh") {
            std::vector<int> idx; std::vector<float> P, N, UV, S;
            if (name == "plymesh") {
                std::string fn = params.oneString("filename", "");
                if (fn.empty()) return fail("plymesh without filename");
                std::string e;
                if (!ReadPlyMesh(resolve(fn), &idx, &P, &N, &UV, &e)) return fail(e);
            } else {
                const std::vector<float> *vi = params.ints("indices");
                const std::vector<float> *vp = params.points("P");
                if (!vi) return fail("Vertex indices \"indices\" not provided with " + name + " shape");
                if (!vp) return fail("Vertex positions \"P\" not provided with " + name + " shape");
                idx.resize(vi->size()); for (size_t i = 0; i < vi->size(); ++i) idx[i] = (int)(*vi)[i];
                P = *vp;
                if (name == "trianglemesh") {   // shapes/triangle.cpp:723-819
                    const std::vector<float> *uv = params.point2s("uv"); if (!uv) uv = params.point2s("st");
                    if (!uv) uv = params.floats("uv");
                    if (!uv) uv = params.floats("st");
                    if (uv) { if (uv->size() / 2 < P.size() / 3) warn("Not enough \"uv\"s for triangle mesh; discarded"); else UV = *uv; }
                    const std::vector<float> *s = params.vectors("S"); if (s && s->size() == P.size()) S = *s;
                    const std::vector<float> *n = params.normals("N"); if (n && n->size() == P.size()) N = *n;
                    if (params.find("alpha", "texture", "float") || params.find("shadowalpha", "texture", "float"))
                        warn("alpha masks are outside the hot-path scope; ignored");
                } else {                        // shapes/loopsubdiv.cpp:399-420
                    int nLevels = params.oneInt("levels", params.oneInt("nlevels", 3));
                    std::vector<int> oi; std::vector<float> oP, oN; std::string e;
                    if (!LoopSubdivide(nLevels, idx, P, &oi, &oP, &oN, &e)) return fail(e);
                    idx.swap(oi); P.swap(oP); N.swap(oN);
                }
            }
            if (idx.size() % 3 != 0) idx.resize(idx.size() - idx.size() % 3);
            return addMeshShape(params, idx, P, N, UV, S);
        } else if (name == "sphere") {          // shapes/sphere.cpp:320-330, shapes/sphere.h:50-59
            ShapeDesc sh;
            sh.kind = kSphere;
            sh.material = materialForShape(params);
            sh.reverseOrientation = gs.reverseOrientation ? 1 : 0;
            sh.transformSwapsHandedness = ctm.swapsHandedness() ? 1 : 0;
            float radius = params.oneFloat("radius", 1.f);
            float zmin = params.oneFloat("zmin", -radius), zmax = params.oneFloat("zmax", radius);
            float phimax = params.oneFloat("phimax", 360.f);
            SphereData &s = sh.sphere;
            s.objectToWorld = ctm.m; s.worldToObject = ctm.inv;
            s.radius = radius;
            s.zMin = clampf(sel_min(zmin, zmax), -radius, radius);
            s.zMax = clampf(sel_max(zmin, zmax), -radius, radius);
            s.thetaMin = std::acos(clampf(sel_min(zmin, zmax) / radius, -1, 1));
            s.thetaMax = std::acos(clampf(sel_max(zmin, zmax) / radius, -1, 1));
            s.phiMax = radians(clampf(phimax, 0, 360));
            sh.areaLight = -1;
            if (!gs.areaLight.empty() && currentObject >= 0)
                warn("Area lights not supported with object instancing (core/api.cpp:1640); the shape is kept without emission");
            else if (!gs.areaLight.empty()) {   // MakeAreaLight, core/api.cpp:782-788 + lights/diffuse.cpp:113-125
                LightDesc l = makeAreaLight();
                l.shape = (int)sc->shapes.size();
                sh.areaLight = (int)sc->lights.size();
                sc->lights.push_back(l);
            }
            pushShape(std::move(sh));
            return true;
        }
        warn("shape \"" + name + "\" is outside the hot-path scope; skipped");
        return true;
    }
    bool doLight(const std::string &name, const ParamList &params) {
        LightDesc l; memset(&l, 0, sizeof(l));
        l.shape = -1;
        float scv[3] = {1, 1, 1};
        params.rgb3("scale", scv);
        if (name == "point") {                  // lights/point.cpp:70-77
            float I[3] = {1, 1, 1};
            params.rgb3("I", I);
            const std::vector<float> *from = params.points("from");
            vec3 P = from && from->size() == 3 ? vec3((*from)[0], (*from)[1], (*from)[2]) : vec3(0, 0, 0);
            Xform l2w = xf_translate(vec3(P.x, P.y, P.z)) * ctm;
            vec3 pw = xf_point(l2w.m, vec3(0, 0, 0));
            l.type = kPointLight;
            l.pos[0] = pw.x; l.pos[1] = pw.y; l.pos[2] = pw.z;
            for (int i = 0; i < 3; ++i) l.I[i] = I[i] * scv[i];
        } else if (name == "distant") {         // lights/distant.cpp:79-88, :44-47
            float L[3] = {1, 1, 1};
            params.rgb3("L", L);
            const std::vector<float> *from = params.points("from"), *to = params.points("to");
            vec3 f = from && from->size() == 3 ? vec3((*from)[0], (*from)[1], (*from)[2]) : vec3(0, 0, 0);
            vec3 t = to && to->size() == 3 ? vec3((*to)[0], (*to)[1], (*to)[2]) : vec3(0, 0, 1);
            vec3 dir = f - t;
            vec3 w = normalize(xf_vector(ctm.m, dir));
            l.type = kDistantLight;
            l.pos[0] = w.x; l.pos[1] = w.y; l.pos[2] = w.z;
            for (int i = 0; i < 3; ++i) l.I[i] = L[i] * scv[i];
        } else if (name == "infinite" || name == "exinfinite") {      // CreateInfiniteLight, lights/infinite.cpp:176-186 (core/api.cpp:748-752)
            float L[3] = {1, 1, 1};
            params.rgb3("L", L);
            for (int i = 0; i < 3; ++i) L[i] = L[i] * scv[i];
            const std::string mapname = params.oneString("mapname", "");
            int w = 1, h = 1; std::vector<float> rgb;
            std::string e;
            if (mapname.empty() || !ReadImageFile(resolve(mapname), &w, &h, &rgb, &e)) {
                // "if (!texels)": a 1x1 map holding L (lights/infinite.cpp:57-61); ReadImage has reported why
                if (!mapname.empty()) warn(e + "; the infinite light uses its constant L");
                w = h = 1; rgb.assign(L, L + 3);
            } else
                for (size_t i = 0; i + 2 < rgb.size(); i += 3) { rgb[i] *= L[0]; rgb[i + 1] *= L[1]; rgb[i + 2] *= L[2]; }
            TextureDesc td;      // MIPMap's defaults: EWA tables, max anisotropy 8, repeat (core/mipmap.h:52-55)
            BuildMipMap(w, h, rgb, 1.f, false, &td, false);
            l.type = kInfiniteLight;
            l.texture = (int32_t)sc->textures.size();
            sc->textures.push_back(std::move(td));
            l.lightToWorld = ctm.m; l.worldToLight = ctm.inv;
            l.I[0] = L[0]; l.I[1] = L[1]; l.I[2] = L[2];
        } else {
            warn("light \"" + name + "\" is outside the hot-path scope; skipped");
            return true;
        }
        sc->lights.push_back(l);
        return true;
    }
    std::string resolve(const std::string &fn) const {
        for (const auto &kv : subst)
            if (!kv.first.empty() && kv.first.back() == '/' && fn.compare(0, kv.first.size(), kv.first) == 0)
                return kv.second + fn.substr(kv.first.size());
        if (!fn.empty() && fn[0] == '/') return fn;
        return baseDir.empty() ? fn : baseDir + "/" + fn;
    }
    bool pushFile(const std::string &path) {
        std::ifstream in(path, std::ios::binary);
        if (!in) return fail("Couldn't open scene file \"" + path + "\"");
        std::ostringstream ss; ss << in.rdbuf();
        Tokenizer t; t.text = ss.str(); t.file = path;
        files.push_back(std::move(t));
        return true;
    }

    void finishOptions() {
        RenderOptions &o = sc->opt;
        // CreateFilm, core/film.cpp:310-349
        if (filmName != "image") warn("Film \"" + filmName + "\" unknown (core/api.cpp:1004-1014); \"image\" used");
        o.filename = filmParams.oneString("filename", "pbrt.exr");
        o.xres = filmParams.oneInt("xresolution", 1280);
        o.yres = filmParams.oneInt("yresolution", 720);
        const std::vector<float> *cr = filmParams.floats("cropwindow");
        if (cr && cr->size() == 4) {
            o.crop[0] = clampf(sel_min((*cr)[0], (*cr)[1]), 0.f, 1.f); o.crop[1] = clampf(sel_max((*cr)[0], (*cr)[1]), 0.f, 1.f);
            o.crop[2] = clampf(sel_min((*cr)[2], (*cr)[3]), 0.f, 1.f); o.crop[3] = clampf(sel_max((*cr)[2], (*cr)[3]), 0.f, 1.f);
        }
        o.filmScale = filmParams.oneFloat("scale", 1.f);
        o.maxSampleLuminance = filmParams.oneFloat("maxsampleluminance", HPRT_INF);
        // MakeFilter: box only (the fork aborts on wider filters)
        if (filterName != "box") warn("pixel filter \"" + filterName + "\" is outside the hot-path scope; box filter used");
        o.filterType = 0;
        o.filterRadius[0] = filterName == "box" ? filterParams.oneFloat("xwidth", 0.5f) : 0.5f;
        o.filterRadius[1] = filterName == "box" ? filterParams.oneFloat("ywidth", 0.5f) : 0.5f;
        // CreatePerspectiveCamera, cameras/perspective.cpp:224-271
        if (cameraName != "perspective") warn("camera \"" + cameraName + "\" is outside the hot-path scope; perspective used");
        o.shutterOpen = cameraParams.oneFloat("shutteropen", 0.f);
        o.shutterClose = cameraParams.oneFloat("shutterclose", 1.f);
        if (o.shutterClose < o.shutterOpen) std::swap(o.shutterClose, o.shutterOpen);
        o.lensRadius = cameraParams.oneFloat("lensradius", 0.f);
        o.focalDistance = cameraParams.oneFloat("focaldistance", 1e6f);
        float frame = cameraParams.oneFloat("frameaspectratio", float(o.xres) / float(o.yres));
        if (frame > 1.f) { o.screenWindow[0] = -frame; o.screenWindow[1] = frame; o.screenWindow[2] = -1.f; o.screenWindow[3] = 1.f; }
        else { o.screenWindow[0] = -1.f; o.screenWindow[1] = 1.f; o.screenWindow[2] = -1.f / frame; o.screenWindow[3] = 1.f / frame; }
        const std::vector<float> *sw = cameraParams.floats("screenwindow");
        if (sw && sw->size() == 4) for (int i = 0; i < 4; ++i) o.screenWindow[i] = (*sw)[i];
        o.fov = cameraParams.oneFloat("fov", 90.f);
        float halffov = cameraParams.oneFloat("halffov", -1.f);
        if (halffov > 0.f) o.fov = 2.f * halffov;
        o.cameraToWorld = cameraToWorld.m; o.worldToCamera = cameraToWorld.inv;
        // CreateHaltonSampler, samplers/halton.cpp:133-139
        o.spp = samplerParams.oneInt("pixelsamples", 16);
        o.samplePixelCenter = samplerParams.oneBool("samplepixelcenter", false) ? 1 : 0;
        // CreatePathIntegrator, integrators/path.cpp:206-229
        o.maxDepth = integratorParams.oneInt("maxdepth", 5);
        o.rrThreshold = integratorParams.oneFloat("rrthreshold", 1.f);
        std::string ls = integratorParams.oneString("lightsamplestrategy", "spatial");
        o.lightStrategy = ls == "uniform" ? kUniform : (ls == "power" ? kPower : kSpatial);
        // CreateBVHAccelerator, accelerators/bvh.cpp:529-535
        o.maxNodePrims = accelParams.oneInt("maxnodeprims", 4);
        o.isectCost = accelParams.oneInt("intersectcost", 8);
        o.travCost = accelParams.oneInt("traversalcost", 1);
        reportUnused(filmParams, "Film", {"diagonal"});
        reportUnused(filterParams, "PixelFilter");
        reportUnused(cameraParams, "Camera");
        reportUnused(samplerParams, "Sampler");
        reportUnused(integratorParams, "Integrator", {"pixelbounds"});
        reportUnused(accelParams, "Accelerator", {"nbDirections", "splitmethod"});
    }

    bool run() {
        std::string tok;
        while (nextToken(&tok)) {
            ParamList pl; std::string name; float f[16];
            if (tok == "AttributeBegin") { gsStack.push_back(gs); xfStack.push_back(ctm); }
            else if (tok == "AttributeEnd") {
                if (gsStack.empty()) { warn("Unmatched AttributeEnd"); continue; }
                gs = gsStack.back(); gsStack.pop_back(); ctm = xfStack.back(); xfStack.pop_back();
            } else if (tok == "TransformBegin") xfStack.push_back(ctm);
            else if (tok == "TransformEnd") { if (!xfStack.empty()) { ctm = xfStack.back(); xfStack.pop_back(); } }
            else if (tok == "Identity") ctm = Xform();
            else if (tok == "Translate") { if (!readFloats(3, f)) return false; ctm = ctm * xf_translate(vec3(f[0], f[1], f[2])); }
            else if (tok == "Scale") { if (!readFloats(3, f)) return false; ctm = ctm * xf_scale(f[0], f[1], f[2]); }
            else if (tok == "Rotate") { if (!readFloats(4, f)) return false; ctm = ctm * xf_rotate(f[0], vec3(f[1], f[2], f[3])); }
            else if (tok == "LookAt") {
                if (!readFloats(9, f)) return false;
                Xform la;
                if (!xf_look_at(vec3(f[0], f[1], f[2]), vec3(f[3], f[4], f[5]), vec3(f[6], f[7], f[8]), &la))
                    warn("LookAt: up vector and viewing direction are collinear; identity used");
                ctm = ctm * la;
            } else if (tok == "Transform" || tok == "ConcatTransform") {
                std::string b;
                if (!nextToken(&b) || b != "[") return fail("expected [ after " + tok);
                if (!readFloats(16, f)) return false;
                if (!nextToken(&b) || b != "]") return fail("expected ] after " + tok);
                mat4 m;   // pbrtTransform transposes, core/api.cpp:1176-1196
                for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m.m[i][j] = f[4 * j + i];
                ctm = (tok == "Transform") ? Xform(m) : ctm * Xform(m);
            } else if (tok == "CoordinateSystem") { if (!readQuoted(&name)) return false; namedCS[name] = ctm; }
            else if (tok == "CoordSysTransform") {
                if (!readQuoted(&name)) return false;
                if (namedCS.count(name)) ctm = namedCS[name]; else warn("Couldn't find named coordinate system \"" + name + "\"");
            } else if (tok == "ReverseOrientation") gs.reverseOrientation = !gs.reverseOrientation;
            else if (tok == "Camera") {
                if (!readQuoted(&cameraName) || !readParams(&cameraParams)) return false;
                cameraToWorld = ctm.inverse();               // core/api.cpp:1303-1309
                namedCS["camera"] = cameraToWorld;
            } else if (tok == "Film") { if (!readQuoted(&filmName) || !readParams(&filmParams)) return false; }
            else if (tok == "PixelFilter") { if (!readQuoted(&filterName) || !readParams(&filterParams)) return false; }
            else if (tok == "Sampler") { if (!readQuoted(&sc->opt.sampler) || !readParams(&samplerParams)) return false; }
            else if (tok == "Accelerator") { if (!readQuoted(&sc->opt.accelerator) || !readParams(&accelParams)) return false; }
            else if (tok == "Integrator") { if (!readQuoted(&sc->opt.integrator) || !readParams(&integratorParams)) return false; }
            else if (tok == "WorldBegin") { inWorld = true; ctm = Xform(); namedCS["world"] = ctm; }
            else if (tok == "WorldEnd") { inWorld = false; }
            else if (tok == "Material") {
                if (!readQuoted(&gs.materialName) || !readParams(&gs.materialParams)) return false;
                gs.namedMaterial.clear();
            } else if (tok == "MakeNamedMaterial") {
                if (!readQuoted(&name) || !readParams(&pl)) return false;
                namedMaterials[name] = std::make_pair(pl.oneString("type", "matte"), pl);
            } else if (tok == "NamedMaterial") { if (!readQuoted(&gs.namedMaterial)) return false; }
            else if (tok == "Texture") {
                std::string ttype, tclass;
                if (!readQuoted(&name) || !readQuoted(&ttype) || !readQuoted(&tclass) || !readParams(&pl)) return false;
                if (tclass == "constant") {                  // textures/constant.cpp
                    TexConst tc; tc.isFloat = (ttype == "float");
                    tc.v[0] = tc.v[1] = tc.v[2] = 1.f;
                    if (tc.isFloat) tc.v[0] = tc.v[1] = tc.v[2] = pl.oneFloat("value", 1.f);
                    else pl.rgb3("value", tc.v);
                    gs.textures[name] = tc;
                } else if (tclass == "imagemap" && (ttype == "spectrum" || ttype == "color")) {
                    // CreateImageSpectrumTexture, textures/imagemap.cpp:140-186
                    const std::string mapping = pl.oneString("mapping", "uv");
                    if (mapping != "uv") { warn("2D texture mapping \"" + mapping + "\" is outside the hot-path scope (uv only)"); continue; }
                    const std::string fn = pl.oneString("filename", "");
                    TextureDesc td;
                    td.su = pl.oneFloat("uscale", 1.f); td.sv = pl.oneFloat("vscale", 1.f);
                    td.du = pl.oneFloat("udelta", 0.f); td.dv = pl.oneFloat("vdelta", 0.f);
                    td.maxAniso = pl.oneFloat("maxanisotropy", 8.f);
                    td.trilinear = pl.oneBool("trilinear", false) ? 1 : 0;
                    const std::string wrap = pl.oneString("wrap", "repeat");
                    td.wrap = wrap == "black" ? kWrapBlack : wrap == "clamp" ? kWrapClamp : kWrapRepeat;
                    const float scale = pl.oneFloat("scale", 1.f);
                    auto ext = [&](const char *e) { size_t n = strlen(e); return fn.size() >= n && strcasecmp(fn.c_str() + fn.size() - n, e) == 0; };
                    const bool gamma = pl.oneBool("gamma", ext(".tga") || ext(".png"));
                    // the texture cache of ImageTexture::GetTexture: same file and parameters, same MIPMap
                    char keyBuf[512];
                    snprintf(keyBuf, sizeof(keyBuf), "|%d|%a|%d|%a|%d|%a|%a|%a|%a", td.trilinear, td.maxAniso, td.wrap, scale, gamma ? 1 : 0, td.su, td.sv, td.du, td.dv);
                    const std::string key = resolve(fn) + keyBuf;
                    int id;
                    auto hit = imageCache.find(key);
                    if (hit != imageCache.end()) id = hit->second;
                    else {
                        int w = 0, h = 0; std::vector<float> rgb; std::string e;
                        if (!ReadImageFile(resolve(fn), &w, &h, &rgb, &e)) {
                            // "Creating a constant grey texture to replace ..." (imagemap.cpp:66-72)
                            warn(e + "; a constant grey texture replaces it");
                            w = h = 1; rgb.assign(3, 0.5f);
                        }
                        BuildMipMap(w, h, rgb, scale, gamma, &td);
                        KeepTextureSource(w, h, rgb, scale, gamma, true, &td);
                        id = (int)sc->textures.size();
                        sc->textures.push_back(std::move(td));
                        imageCache[key] = id;
                    }
                    TexConst tc; tc.isFloat = false; tc.v[0] = tc.v[1] = tc.v[2] = 0.f; tc.image = id;
                    gs.textures[name] = tc;
                    reportUnused(pl, "Texture \"" + name + "\"");
                } else warn("texture class \"" + tclass + "\" is outside the hot-path scope (constant, spectrum imagemap)");
            } else if (tok == "AreaLightSource") { if (!readQuoted(&gs.areaLight) || !readParams(&gs.areaLightParams)) return false; }
            else if (tok == "LightSource") {
                if (!readQuoted(&name) || !readParams(&pl) || !doLight(name, pl)) return false;
                if (name == "point" || name == "distant" || name == "infinite" || name == "exinfinite") reportUnused(pl, "LightSource \"" + name + "\"", {"nsamples", "samples"});
            } else if (tok == "Shape") {
                if (!readQuoted(&name) || !readParams(&pl) || !doShape(name, pl)) return false;
                if (name == "trianglemesh" || name == "loopsubdiv" || name == "plymesh" || name == "sphere") reportUnused(pl, "Shape \"" + name + "\"");
            }
            else if (tok == "Include") {
                if (!readQuoted(&name)) return false;
                if (!pushFile(resolve(name))) return false;
            } else if (tok == "ObjectBegin") {               // core/api.cpp:1752-1761
                if (!readQuoted(&name)) return false;
                gsStack.push_back(gs); xfStack.push_back(ctm);
                if (currentObject >= 0) return fail("ObjectBegin called inside of instance definition");
                currentObject = (int)sc->nObjects++;
                objectByName[name] = currentObject;          // a later definition replaces an earlier one of the same name
            } else if (tok == "ObjectEnd") {                 // core/api.cpp:1765-1774
                if (currentObject < 0) return fail("ObjectEnd called outside of instance definition");
                currentObject = -1;
                if (!gsStack.empty()) { gs = gsStack.back(); gsStack.pop_back(); ctm = xfStack.back(); xfStack.pop_back(); }
            } else if (tok == "ObjectInstance") {            // core/api.cpp:1778-1820
                if (!readQuoted(&name)) return false;
                if (currentObject >= 0) return fail("ObjectInstance can't be called inside instance definition");
                if (!objectByName.count(name)) return fail("Unable to find instance named \"" + name + "\"");
                const int obj = objectByName[name];
                bool any = false;
                for (const ShapeDesc &sh : sc->shapes) if (sh.object == obj && sh.nPrims() > 0) { any = true; break; }
                if (!any) continue;                          // "if (in.empty()) return;"
                InstanceDesc in; in.object = obj; in.instanceToWorld = ctm.m; in.worldToInstance = ctm.inv;
                sc->top.push_back(TopItem{1, (uint32_t)sc->instances.size()});
                sc->instances.push_back(in);
            }
            else if (tok == "MediumInterface" || tok == "MakeNamedMedium") {
                warn(tok + " ignored (media are outside the hot-path scope)");
                std::string t2; while (nextToken(&t2)) { if (!quoted(t2) && t2 != "[" && t2 != "]") { float d; if (!parseNumber(t2, &d)) { unget(t2); break; } } }
            } else return fail("Unexpected token: " + tok);
        }
        finishOptions();
        return true;
    }
};

}  // namespace

bool ParsePbrtFile(const std::string &path, const std::map<std::string, std::string> &subst, SceneModel *sc,
                   std::string *err) {
    Frontend fe; fe.sc = sc; fe.subst = subst;
    size_t slash = path.find_last_of('/');
    fe.baseDir = slash == std::string::npos ? "" : path.substr(0, slash);
    if (!fe.pushFile(path) || !fe.run()) { *err = fe.err; return false; }
    return true;
}
bool ParsePbrtString(const std::string &text, const std::string &baseDir,
                     const std::map<std::string, std::string> &subst, SceneModel *sc, std::string *err) {
    Frontend fe; fe.sc = sc; fe.baseDir = baseDir; fe.subst = subst;
    Tokenizer t; t.text = text; t.file = "<string>";
    fe.files.push_back(std::move(t));
    if (!fe.run()) { *err = fe.err; return false; }
    return true;
}

}  // namespace hprt
