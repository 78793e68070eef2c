// accelerators/hprt.cpp in the reference tree — see hprt_accel.h.  NOT compiled in this repository.
//
// What the constructor and Device() do, and which reference members they read:
//   primitives[i]                       GeometricPrimitive (core/primitive.h:136-140: shape, material, areaLight — private: the
//                                       friend declarations of integration/README.md) or TransformedPrimitive (:103-128)
//   GeometricPrimitive::shape           Triangle (shapes/triangle.h:80-146: mesh, v) -> TriangleMesh (:54-75), or Sphere
//                                       (shapes/sphere.h:50-81: radius, zMin, zMax, thetaMin, thetaMax, phiMax)
//   Shape::ObjectToWorld / WorldToObject, reverseOrientation, transformSwapsHandedness      core/shape.h:76-80
//   GeometricPrimitive::material        Matte / Plastic / Mirror / Substrate / Metal / Glass / UberMaterial: their texture members
//                                       evaluated as constants, or as ImageTexture<RGBSpectrum, Spectrum> -> MIPMap pyramid
//   GeometricPrimitive::areaLight       DiffuseAreaLight (lights/diffuse.h:69-77: Lemit, twoSided); its index in Scene::lights
//   Primitive::WorldBound()             per primitive, pbrt's own arithmetic: the aggregate is built from exactly the bounds
//                                       BVHAccel's constructor would see (accelerators/bvh.cpp:163-166)
#include "accelerators/hprt.h"

#include "interaction.h"
#include "lights/diffuse.h"
#include "lights/distant.h"
#include "lights/infinite.h"
#include "lights/point.h"
#include "materials/glass.h"
#include "materials/matte.h"
#include "materials/metal.h"
#include "materials/mirror.h"
#include "materials/plastic.h"
#include "materials/substrate.h"
#include "materials/uber.h"
#include "paramset.h"
#include "shapes/sphere.h"
#include "shapes/triangle.h"
#include "textures/constant.h"
#include "textures/imagemap.h"

namespace pbrt {

const HprtAccel *HprtAccel::topLevel = nullptr;

// ---------------------------------------------------------------------------------------------------------------------------
// HprtSceneWalk: one pass over a primitive vector (and, behind TransformedPrimitives, over the object definitions' vectors),
// collecting the bridge's plain structs.  Pointers into pbrt's own arrays are borrowed; everything synthesised here (runs,
// bounds, descriptors, texture level tables) is owned by the walk, which lives as long as the device scene is being created.
// ---------------------------------------------------------------------------------------------------------------------------
class HprtSceneWalk {
  public:
    std::vector<HprtBridgeMesh> meshes;
    std::vector<HprtBridgeSphere> spheres;
    std::vector<HprtBridgeInstance> instances;
    std::vector<HprtMaterialDesc> materials;
    std::vector<HprtLightDesc> lights;
    std::vector<HprtTextureDesc> textures;
    struct PrimList { std::vector<HprtBridgeRun> runs; std::vector<float> bounds; };
    PrimList top;
    std::vector<std::unique_ptr<PrimList>> objectLists;
    std::vector<HprtBridgeObject> objects;

    HprtSceneWalk(const std::vector<std::shared_ptr<Light>> &sceneLights) : sceneLights(sceneLights) {}

    // ---- lights: Scene::lights order (UniformSampleOneLight indexes it, core/integrator.cpp:94-99) ----
    void Lights() {
        for (const std::shared_ptr<Light> &l : sceneLights) {
            HprtLightDesc d; memset(&d, 0, sizeof(d));
            d.shape = -1; d.texture = -1;
            Float rgb[3];
            if (const PointLight *pl = dynamic_cast<const PointLight *>(l.get())) {                 // lights/point.h:67-70
                d.type = 0; Copy(pl->pLight, d.pos); pl->I.ToRGB(rgb); Copy3(rgb, d.I);
            } else if (const DistantLight *dl = dynamic_cast<const DistantLight *>(l.get())) {      // lights/distant.h:68-73
                d.type = 1; Copy(dl->wLight, d.pos); dl->L.ToRGB(rgb); Copy3(rgb, d.I);
            } else if (const DiffuseAreaLight *al = dynamic_cast<const DiffuseAreaLight *>(l.get())) {   // lights/diffuse.h:69-77
                d.type = 2; al->Lemit.ToRGB(rgb); Copy3(rgb, d.I); d.two_sided = al->twoSided ? 1 : 0;
                areaLightIndex[al] = (int)lights.size();
            } else if (const InfiniteAreaLight *il = dynamic_cast<const InfiniteAreaLight *>(l.get())) {   // lights/infinite.h:55-83
                d.type = 3; d.I[0] = d.I[1] = d.I[2] = 1.f;
                d.texture = TextureOf(il->Lmap.get(), 1.f, 1.f, 0.f, 0.f);                           // texels already times L * scale
                CopyMatrix(il->LightToWorld.GetMatrix(), d.light_to_world); CopyMatrix(il->WorldToLight.GetMatrix(), d.world_to_light);
            } else
                Error("hprt: light type outside the hot path's scope (point, distant, diffuse area, infinite)");
            lights.push_back(d);
        }
    }

    // ---- one primitive vector -> runs + per-primitive world bounds ----
    void List(const std::vector<std::shared_ptr<Primitive>> &prims, PrimList *out, bool topLevel) {
        const TriangleMesh *runMesh = nullptr;
        for (const std::shared_ptr<Primitive> &prim : prims) {
            const Bounds3f b = prim->WorldBound();                                  // Triangle::WorldBound / Sphere via Shape::WorldBound /
            const float b6[6] = {b.pMin.x, b.pMin.y, b.pMin.z, b.pMax.x, b.pMax.y, b.pMax.z};   // TransformedPrimitive::WorldBound
            out->bounds.insert(out->bounds.end(), b6, b6 + 6);
            if (const TransformedPrimitive *tp = dynamic_cast<const TransformedPrimitive *>(prim.get())) {
                CHECK(topLevel);                                                    // core/api.cpp:1786-1789 forbids nesting
                runMesh = nullptr;
                HprtBridgeInstance in;
                in.object = ObjectOf(tp->primitive);
                // PrimitiveToWorld's start transform: instances are static in the hot path's scope
                CopyMatrix(tp->PrimitiveToWorld.startTransform->GetMatrix(), in.instance_to_world);
                CopyMatrix(tp->PrimitiveToWorld.startTransform->GetInverseMatrix(), in.world_to_instance);
                out->runs.push_back(HprtBridgeRun{2, (int32_t)instances.size()});
                instances.push_back(in);
                continue;
            }
            const GeometricPrimitive *gp = dynamic_cast<const GeometricPrimitive *>(prim.get());
            CHECK(gp != nullptr);                                                   // nothing else is ever in a primitive vector
            if (const Triangle *tri = dynamic_cast<const Triangle *>(gp->shape.get())) {
                // the Triangles of one TriangleMesh are consecutive and in face order (CreateTriangleMesh, shapes/triangle.cpp:
                // 94-112; pbrtShape pushes them back in that order, core/api.cpp:1609-1648): one run per mesh
                if (tri->mesh.get() != runMesh) {
                    runMesh = tri->mesh.get();
                    const TriangleMesh &m = *runMesh;
                    HprtBridgeMesh bm; memset(&bm, 0, sizeof(bm));
                    bm.n_triangles = m.nTriangles; bm.n_vertices = m.nVertices;
                    bm.vertex_indices = m.vertexIndices.data();
                    bm.p = reinterpret_cast<const float *>(m.p.get());             // Point3f = {Float x, y, z} (core/geometry.h)
                    bm.n = reinterpret_cast<const float *>(m.n.get());
                    bm.s = reinterpret_cast<const float *>(m.s.get());
                    bm.uv = reinterpret_cast<const float *>(m.uv.get());
                    bm.reverse_orientation = tri->reverseOrientation ? 1 : 0;
                    bm.transform_swaps_handedness = tri->transformSwapsHandedness ? 1 : 0;
                    bm.material = MaterialOf(gp->material.get());
                    bm.first_area_light = gp->areaLight ? AreaLightOf(gp->areaLight.get()) : -1;
                    if (m.alphaMask || m.shadowAlphaMask) Error("hprt: triangle alpha masks are outside the hot path's scope");
                    out->runs.push_back(HprtBridgeRun{0, (int32_t)meshes.size()});
                    meshes.push_back(bm);
                }
            } else if (const Sphere *sp = dynamic_cast<const Sphere *>(gp->shape.get())) {
                runMesh = nullptr;
                HprtBridgeSphere bs; memset(&bs, 0, sizeof(bs));
                CopyMatrix(sp->ObjectToWorld->GetMatrix(), bs.object_to_world); CopyMatrix(sp->WorldToObject->GetMatrix(), bs.world_to_object);
                bs.radius = sp->radius; bs.z_min = sp->zMin; bs.z_max = sp->zMax;
                bs.theta_min = sp->thetaMin; bs.theta_max = sp->thetaMax; bs.phi_max = sp->phiMax;
                bs.reverse_orientation = sp->reverseOrientation ? 1 : 0; bs.transform_swaps_handedness = sp->transformSwapsHandedness ? 1 : 0;
                bs.material = MaterialOf(gp->material.get());
                bs.area_light = gp->areaLight ? AreaLightOf(gp->areaLight.get()) : -1;
                out->runs.push_back(HprtBridgeRun{1, (int32_t)spheres.size()});
                spheres.push_back(bs);
            } else
                Error("hprt: shape type outside the hot path's scope (triangle meshes and spheres)");
        }
    }

    void Fill(HprtBridgeScene *sc, int lightStrategy, int maxPrimsInNode, int isectCost, int travCost) {
        for (size_t o = 0; o < objectLists.size(); ++o)
            objects.push_back(HprtBridgeObject{objectLists[o]->runs.data(), (uint32_t)objectLists[o]->runs.size(), objectLists[o]->bounds.data()});
        memset(sc, 0, sizeof(*sc));
        sc->meshes = meshes.data(); sc->n_meshes = (uint32_t)meshes.size();
        sc->spheres = spheres.data(); sc->n_spheres = (uint32_t)spheres.size();
        sc->runs = top.runs.data(); sc->n_runs = (uint32_t)top.runs.size(); sc->prim_bounds = top.bounds.data();
        sc->objects = objects.data(); sc->n_objects = (uint32_t)objects.size();
        sc->instances = instances.data(); sc->n_instances = (uint32_t)instances.size();
        sc->materials = materials.data(); sc->n_materials = (uint32_t)materials.size();
        sc->lights = lights.data(); sc->n_lights = (uint32_t)lights.size();
        sc->textures = textures.data(); sc->n_textures = (uint32_t)textures.size();
        sc->light_strategy = lightStrategy;
        sc->max_node_prims = maxPrimsInNode; sc->isect_cost = isectCost; sc->trav_cost = travCost;
    }

  private:
    const std::vector<std::shared_ptr<Light>> &sceneLights;
    std::map<const AreaLight *, int> areaLightIndex;
    std::map<const Material *, int> materialIndex;
    std::map<const Primitive *, int> objectIndex;
    std::map<const void *, int> textureIndex;
    std::vector<std::unique_ptr<std::vector<HprtTextureLevel>>> levelTables;
    std::vector<std::unique_ptr<std::vector<float>>> texelStore;

    static void Copy(const Point3f &p, float o[3]) { o[0] = p.x; o[1] = p.y; o[2] = p.z; }
    static void Copy(const Vector3f &p, float o[3]) { o[0] = p.x; o[1] = p.y; o[2] = p.z; }
    static void Copy3(const Float v[3], float o[3]) { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; }
    static void CopyMatrix(const Matrix4x4 &m, float o[16]) { for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) o[4 * i + j] = m.m[i][j]; }

    // The DiffuseAreaLights of a mesh's faces are consecutive in Scene::lights (core/api.cpp:1630-1636 pushes one per Triangle, in
    // face order): the first face's index stands for the mesh
    int AreaLightOf(const AreaLight *al) {
        auto it = areaLightIndex.find(al);
        CHECK(it != areaLightIndex.end());
        return it->second;
    }

    // An object definition behind a TransformedPrimitive: the HprtAccel pbrtObjectInstance built over its primitives, or — for
    // an object with a single primitive — that primitive itself (core/api.cpp:1798-1806)
    int ObjectOf(const std::shared_ptr<Primitive> &p) {
        auto it = objectIndex.find(p.get());
        if (it != objectIndex.end()) return it->second;
        const int id = (int)objectLists.size();
        objectIndex[p.get()] = id;
        objectLists.emplace_back(new PrimList());
        if (const HprtAccel *inner = dynamic_cast<const HprtAccel *>(p.get())) List(inner->primitives, objectLists[id].get(), false);
        else { std::vector<std::shared_ptr<Primitive>> one{p}; List(one, objectLists[id].get(), false); }
        return id;
    }

    // ---- materials: the Texture members evaluated as constants (ConstantTexture::Evaluate ignores its argument) or bound as images ----
    static bool Constant(const std::shared_ptr<Texture<Spectrum>> &t, float out[3]) {
        if (!t) return false;
        if (!dynamic_cast<const ConstantTexture<Spectrum> *>(t.get())) return false;
        Float rgb[3]; t->Evaluate(SurfaceInteraction()).ToRGB(rgb);
        out[0] = rgb[0]; out[1] = rgb[1]; out[2] = rgb[2];
        return true;
    }
    static float ConstantFloat(const std::shared_ptr<Texture<Float>> &t, float def) {
        if (!t) return def;
        if (!dynamic_cast<const ConstantTexture<Float> *>(t.get())) { Error("hprt: float image textures are outside the hot path's scope"); return def; }
        return t->Evaluate(SurfaceInteraction());
    }
    // constant -> out[], returns -1; ImageTexture<RGBSpectrum, Spectrum> -> its index in `textures`
    int Spectral(const std::shared_ptr<Texture<Spectrum>> &t, float out[3]) {
        if (Constant(t, out)) return -1;
        if (const ImageTexture<RGBSpectrum, Spectrum> *it = dynamic_cast<const ImageTexture<RGBSpectrum, Spectrum> *>(t.get())) {
            const UVMapping2D *uv = dynamic_cast<const UVMapping2D *>(it->mapping.get());      // textures/imagemap.h:92-94, core/texture.h:71-80
            if (!uv) { Error("hprt: only UVMapping2D is in the hot path's scope"); return -1; }
            return TextureOf(it->mipmap, uv->su, uv->sv, uv->du, uv->dv);
        }
        Error("hprt: texture class outside the hot path's scope (constant, spectrum imagemap)");
        return -1;
    }
    // MIPMap<RGBSpectrum> (core/mipmap.h:53-110): pyramid levels read out of their BlockedArrays into row-major RGB floats (row 0 = t 0)
    int TextureOf(const MIPMap<RGBSpectrum> *mip, float su, float sv, float du, float dv) {
        auto found = textureIndex.find(mip);
        if (found != textureIndex.end()) return found->second;
        HprtTextureDesc d; memset(&d, 0, sizeof(d));
        levelTables.emplace_back(new std::vector<HprtTextureLevel>());
        for (int l = 0; l < mip->Levels(); ++l) {
            const BlockedArray<RGBSpectrum> &lev = *mip->pyramid[l];
            texelStore.emplace_back(new std::vector<float>(3 * (size_t)lev.uSize() * lev.vSize()));
            std::vector<float> &px = *texelStore.back();
            for (int t = 0; t < lev.vSize(); ++t)
                for (int s = 0; s < lev.uSize(); ++s) {
                    Float rgb[3]; lev(s, t).ToRGB(rgb);
                    float *o = &px[3 * ((size_t)t * lev.uSize() + s)];
                    o[0] = rgb[0]; o[1] = rgb[1]; o[2] = rgb[2];
                }
            levelTables.back()->push_back(HprtTextureLevel{lev.uSize(), lev.vSize(), px.data()});
        }
        d.levels = levelTables.back()->data(); d.n_levels = (uint32_t)levelTables.back()->size();
        d.trilinear = mip->doTrilinear ? 1 : 0; d.max_anisotropy = mip->maxAnisotropy;
        d.wrap = mip->wrapMode == ImageWrap::Repeat ? 0 : mip->wrapMode == ImageWrap::Black ? 1 : 2;
        d.su = su; d.sv = sv; d.du = du; d.dv = dv;
        d.weight_lut = MIPMap<RGBSpectrum>::weightLut;                                              // core/mipmap.h:108
        const int id = (int)textures.size();
        textures.push_back(d);
        textureIndex[mip] = id;
        return id;
    }

    int MaterialOf(const Material *m) {
        auto found = materialIndex.find(m);
        if (found != materialIndex.end()) return found->second;
        HprtMaterialDesc d; memset(&d, 0, sizeof(d));
        d.kd_texture = d.ks_texture = d.opacity_texture = -1;
        d.opacity[0] = d.opacity[1] = d.opacity[2] = 1.f; d.eta = 1.5f; d.remap_roughness = 1;
        if (const MatteMaterial *mm = dynamic_cast<const MatteMaterial *>(m)) {                     // materials/matte.h:59-62
            d.type = 0; d.kd_texture = Spectral(mm->Kd, d.Kd); d.sigma = ConstantFloat(mm->sigma, 0.f);
            if (mm->bumpMap) Error("hprt: bump maps are outside the hot path's scope");
        } else if (const PlasticMaterial *pm = dynamic_cast<const PlasticMaterial *>(m)) {          // materials/plastic.h:65-69
            d.type = 1; d.kd_texture = Spectral(pm->Kd, d.Kd); d.ks_texture = Spectral(pm->Ks, d.Ks);
            d.roughness = ConstantFloat(pm->roughness, .1f); d.remap_roughness = pm->remapRoughness ? 1 : 0;
        } else if (const MirrorMaterial *rm = dynamic_cast<const MirrorMaterial *>(m)) {            // materials/mirror.h:60-63: Kr travels in Ks
            d.type = 2; d.ks_texture = Spectral(rm->Kr, d.Ks);
        } else if (const SubstrateMaterial *sm = dynamic_cast<const SubstrateMaterial *>(m)) {      // materials/substrate.h:67-72
            d.type = 3; d.kd_texture = Spectral(sm->Kd, d.Kd); d.ks_texture = Spectral(sm->Ks, d.Ks);
            d.roughness = ConstantFloat(sm->nu, .1f); d.sigma = ConstantFloat(sm->nv, .1f); d.remap_roughness = sm->remapRoughness ? 1 : 0;
        } else if (const MetalMaterial *tm = dynamic_cast<const MetalMaterial *>(m)) {              // materials/metal.h:63-68: Kd = eta, Ks = k
            d.type = 4;
            if (!Constant(tm->eta, d.Kd) || !Constant(tm->k, d.Ks)) Error("hprt: metal eta / k must be constant spectra");
            const float r = ConstantFloat(tm->roughness, .01f);
            d.roughness = tm->uRoughness ? ConstantFloat(tm->uRoughness, r) : r; d.sigma = tm->vRoughness ? ConstantFloat(tm->vRoughness, r) : r;
            d.remap_roughness = tm->remapRoughness ? 1 : 0;
        } else if (const GlassMaterial *gm = dynamic_cast<const GlassMaterial *>(m)) {              // materials/glass.h:69-75: Kd = Kt, Ks = Kr, roughness = index
            d.type = 5; d.ks_texture = Spectral(gm->Kr, d.Ks); d.kd_texture = Spectral(gm->Kt, d.Kd);
            d.roughness = ConstantFloat(gm->index, 1.5f);
            if (ConstantFloat(gm->uRoughness, 0.f) != 0.f || ConstantFloat(gm->vRoughness, 0.f) != 0.f)
                Warning("hprt: rough glass (MicrofacetTransmission) is outside the hot path's scope; rendered as smooth glass");
        } else if (const UberMaterial *um = dynamic_cast<const UberMaterial *>(m)) {                // materials/uber.h:77-82
            d.type = 6; d.kd_texture = Spectral(um->Kd, d.Kd); d.ks_texture = Spectral(um->Ks, d.Ks);
            if (!Constant(um->Kr, d.Kr) || !Constant(um->Kt, d.Kt)) Error("hprt: uber Kr / Kt must be constant spectra");
            d.opacity_texture = Spectral(um->opacity, d.opacity);
            const float r = ConstantFloat(um->roughness, .1f);
            d.roughness = um->roughnessu ? ConstantFloat(um->roughnessu, r) : r;                    // materials/uber.cpp:76-85
            d.sigma = um->roughnessv ? ConstantFloat(um->roughnessv, d.roughness) : d.roughness;
            d.eta = ConstantFloat(um->eta, 1.5f); d.remap_roughness = um->remapRoughness ? 1 : 0;
        } else
            Error("hprt: material outside the hot path's scope (matte, plastic, mirror, substrate, metal, glass, uber); rendered as matte"), d.type = 0,
                d.Kd[0] = d.Kd[1] = d.Kd[2] = .5f;
        const int id = (int)materials.size();
        materials.push_back(d);
        materialIndex[m] = id;
        return id;
    }
};

// ---------------------------------------------------------------------------------------------------------------------------
HprtAccel::HprtAccel(std::vector<std::shared_ptr<Primitive>> p, int maxPrimsInNode, int isectCost, int travCost)
    : primitives(std::move(p)), maxPrimsInNode(std::min(255, maxPrimsInNode)), isectCost(isectCost), travCost(travCost) {
    for (const std::shared_ptr<Primitive> &prim : primitives) bounds = Union(bounds, prim->WorldBound());      // bvh.cpp:187-189's root bounds
    topLevel = this;      // MakeScene() constructs the scene's aggregate after every object definition's (core/api.cpp:1883-1886)
}

HprtAccel::~HprtAccel() {
    if (device) hprt_bridge_accel_destroy(device);
    if (topLevel == this) topLevel = nullptr;
}

HprtBridgeAccel *HprtAccel::Device(const std::vector<std::shared_ptr<Light>> &lights, int lightStrategy) const {
    std::lock_guard<std::mutex> lock(deviceMutex);
    if (device) return device;
    HprtSceneWalk walk(lights);
    walk.Lights();                          // first: shapes look their DiffuseAreaLight's index up
    walk.List(primitives, &walk.top, true);
    HprtBridgeScene sc;
    walk.Fill(&sc, lightStrategy, maxPrimsInNode, isectCost, travCost);
    HprtBridgeAccel *a = nullptr;
    if (hprt_bridge_accel_build(&sc, &a) != HPRT_OK || hprt_bridge_accel_upload(a, -1) != HPRT_OK) {
        Error("hprt: %s", hprt_last_error());
        if (a) hprt_bridge_accel_destroy(a);
        return nullptr;
    }
    device = a;                             // the library has copied everything to HBM: the walk's arrays may go
    return device;
}

// Aggregate::Intersect through the batch ABI with one ray.  A contract shim (see the header): ~tens of microseconds per call.
bool HprtAccel::Intersect(const Ray &ray, SurfaceInteraction *isect) const {
    HprtBridgeAccel *a = Device(std::vector<std::shared_ptr<Light>>(), 0);
    if (!a) return false;
    const float o[3] = {ray.o.x, ray.o.y, ray.o.z}, d[3] = {ray.d.x, ray.d.y, ray.d.z}, tmax = ray.tMax;
    float t, bary[3]; int32_t prim, inst;
    if (hprt_intersect_instanced(hprt_bridge_accel_scene(a), 1, o, d, &tmax, &t, &prim, &inst, bary, nullptr) != HPRT_OK || prim < 0) return false;
    // The ABI returns the ORDERED primitive index and the t / barycentrics the reference computes (shapes/triangle.cpp:264-269).
    // The SurfaceInteraction is the reference's own: ask the primitive that was hit again, with tMax just past the hit — its
    // Intersect recomputes the same t and fills *isect (and shrinks ray.tMax) exactly as BVHAccel::Intersect's leaf loop does.
    const HprtSceneDesc *desc = hprt_bridge_accel_desc(a);
    const uint32_t creation = desc->prim_order[prim < (int32_t)desc->n_prims ? prim : 0];
    if (inst < 0 && prim < (int32_t)desc->n_prims) return primitives[creation]->Intersect(ray, isect);
    // a hit inside an object instance: the TransformedPrimitive at its top-level position does the rest (core/primitive.cpp:77-93)
    for (const std::shared_ptr<Primitive> &p : primitives)
        if (dynamic_cast<const TransformedPrimitive *>(p.get()) && p->Intersect(ray, isect)) return true;
    return false;
}

bool HprtAccel::IntersectP(const Ray &ray) const {
    HprtBridgeAccel *a = Device(std::vector<std::shared_ptr<Light>>(), 0);
    if (!a) return false;
    const float o[3] = {ray.o.x, ray.o.y, ray.o.z}, d[3] = {ray.d.x, ray.d.y, ray.d.z}, tmax = ray.tMax;
    uint8_t occ = 0;
    return hprt_occluded(hprt_bridge_accel_scene(a), 1, o, d, &tmax, &occ, nullptr) == HPRT_OK && occ != 0;
}

std::shared_ptr<HprtAccel> CreateHprtAccelerator(std::vector<std::shared_ptr<Primitive>> prims, const ParamSet &ps) {
    // the parameters CreateBVHAccelerator reads (accelerators/bvh.cpp:529-535), same names and defaults
    const int maxPrimsInNode = ps.FindOneInt("maxnodeprims", 4);
    const int isectCost = ps.FindOneInt("intersectcost", 8);
    const int travCost = ps.FindOneInt("traversalcost", 1);
    return std::make_shared<HprtAccel>(std::move(prims), maxPrimsInNode, isectCost, travCost);
}

}  // namespace pbrt
