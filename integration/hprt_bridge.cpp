// hprt_bridge — see hprt_bridge.h.  Plain C++17 over include/hprt.h only; built by tests/test_integration_bridge.py
//   g++ -O2 -std=c++17 -fPIC -shared -Wall -Werror integration/hprt_bridge.cpp -o libhprt_bridge.so -L<lib> -lhprt
#include "hprt_bridge.h"

#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace {

struct BvhArrays {
    HprtBvh *bvh = nullptr;
    std::vector<uint8_t> nodes; std::vector<uint32_t> order; uint32_t nNodes = 0, nPrims = 0;
    ~BvhArrays() { if (bvh) hprt_bvh_destroy(bvh); }
};

// BVHAccel::BVHAccel over primitives whose WorldBound()s are given (accelerators/bvh.cpp:155-185): the library's host builder
// restates iterativeBuild / flattenBVHTree, so the arrays are the ones BVHAccel would hold
int BuildAggregate(const float *bounds6, size_t n, const HprtBridgeScene &in, BvhArrays *out) {
    std::vector<float> lo(3 * n), hi(3 * n);
    for (size_t i = 0; i < n; ++i) { memcpy(&lo[3 * i], bounds6 + 6 * i, 12); memcpy(&hi[3 * i], bounds6 + 6 * i + 3, 12); }
    int rc = hprt_bvh_build_from_bounds(n, lo.data(), hi.data(), in.max_node_prims, in.isect_cost, in.trav_cost, &out->bvh);
    if (rc != HPRT_OK) return rc;
    uint32_t info[4];
    if ((rc = hprt_bvh_info(out->bvh, info, nullptr)) != HPRT_OK) return rc;
    out->nNodes = info[0]; out->nPrims = info[1];
    out->nodes.resize((size_t)out->nNodes * 32); out->order.resize(out->nPrims);
    return hprt_bvh_copy(out->bvh, out->nodes.data(), out->order.data());
}

}  // namespace

struct HprtBridgeAccel {
    BvhArrays top;
    std::vector<std::unique_ptr<BvhArrays>> objectBvh;
    std::vector<HprtShapeDesc> shapes;
    std::vector<HprtLightDesc> lights;
    std::vector<HprtObjectDesc> objects;
    std::vector<HprtInstanceDesc> instances;
    std::vector<HprtTopItem> topItems;
    HprtSceneDesc desc;
    HprtScene *scene = nullptr;
    float bounds[6] = {0, 0, 0, 0, 0, 0};
    ~HprtBridgeAccel() { if (scene) hprt_scene_destroy(scene); }
};

extern "C" {

int hprt_bridge_accel_build(const HprtBridgeScene *in, HprtBridgeAccel **out) {
    if (!in || !out) return HPRT_E_INVALID;
    std::unique_ptr<HprtBridgeAccel> a(new HprtBridgeAccel());
    a->lights.assign(in->lights, in->lights + in->n_lights);
    // one HprtShapeDesc per run of kind 0 / 1; an object's shapes are consecutive (HprtObjectDesc::first_shape / n_shapes)
    auto addShape = [&](const HprtBridgeRun &r) -> int {
        HprtShapeDesc s; memset(&s, 0, sizeof(s));
        if (r.kind == 0) {
            if (r.index < 0 || (uint32_t)r.index >= in->n_meshes) return HPRT_E_INVALID;
            const HprtBridgeMesh &m = in->meshes[r.index];
            s.kind = 0; s.material = m.material; s.area_light = m.first_area_light;
            s.reverse_orientation = m.reverse_orientation; s.transform_swaps_handedness = m.transform_swaps_handedness;
            s.n_tris = (uint32_t)m.n_triangles; s.n_verts = (uint32_t)m.n_vertices;
            s.indices = m.vertex_indices; s.P = m.p; s.N = m.n; s.UV = m.uv; s.S = m.s;
            // every face of an emissive mesh has its own DiffuseAreaLight (core/api.cpp:1609-1636)
            if (m.first_area_light >= 0)
                for (int32_t t = 0; t < m.n_triangles; ++t) {
                    if ((uint32_t)(m.first_area_light + t) >= in->n_lights) return HPRT_E_INVALID;
                    a->lights[m.first_area_light + t].shape = (int32_t)a->shapes.size();
                }
        } else if (r.kind == 1) {
            if (r.index < 0 || (uint32_t)r.index >= in->n_spheres) return HPRT_E_INVALID;
            const HprtBridgeSphere &sp = in->spheres[r.index];
            s.kind = 1; s.material = sp.material; s.area_light = sp.area_light;
            s.reverse_orientation = sp.reverse_orientation; s.transform_swaps_handedness = sp.transform_swaps_handedness;
            memcpy(s.object_to_world, sp.object_to_world, 64); memcpy(s.world_to_object, sp.world_to_object, 64);
            s.radius = sp.radius; s.z_min = sp.z_min; s.z_max = sp.z_max; s.theta_min = sp.theta_min; s.theta_max = sp.theta_max; s.phi_max = sp.phi_max;
            if (sp.area_light >= 0) { if ((uint32_t)sp.area_light >= in->n_lights) return HPRT_E_INVALID; a->lights[sp.area_light].shape = (int32_t)a->shapes.size(); }
        } else return HPRT_E_INVALID;
        a->shapes.push_back(s);
        return HPRT_OK;
    };
    auto primsOf = [&](const HprtBridgeRun &r) -> size_t { return r.kind == 0 ? (size_t)in->meshes[r.index].n_triangles : 1u; };
    // ---- object definitions: their shapes, their aggregates ----
    for (uint32_t o = 0; o < in->n_objects; ++o) {
        const HprtBridgeObject &ob = in->objects[o];
        HprtObjectDesc od; memset(&od, 0, sizeof(od));
        od.first_shape = (uint32_t)a->shapes.size();
        size_t n = 0;
        for (uint32_t k = 0; k < ob.n_runs; ++k) {
            if (ob.runs[k].kind == 2) return HPRT_E_INVALID;      // "ObjectInstance can't be called inside instance definition" (core/api.cpp:1781)
            if (int rc = addShape(ob.runs[k])) return rc;
            n += primsOf(ob.runs[k]);
        }
        od.n_shapes = (uint32_t)a->shapes.size() - od.first_shape;
        a->objectBvh.emplace_back(new BvhArrays());
        BvhArrays &b = *a->objectBvh.back();
        if (int rc = BuildAggregate(ob.prim_bounds, n, *in, &b)) return rc;
        od.nodes = b.nodes.data(); od.n_nodes = b.nNodes; od.prim_order = b.order.data(); od.n_prims = b.nPrims;
        a->objects.push_back(od);
    }
    // ---- the top level, in the order of the primitive vector ----
    size_t nTop = 0;
    for (uint32_t k = 0; k < in->n_runs; ++k) {
        const HprtBridgeRun &r = in->runs[k];
        if (r.kind == 2) {
            if (r.index < 0 || (uint32_t)r.index >= in->n_instances) return HPRT_E_INVALID;
            const HprtBridgeInstance &bi = in->instances[r.index];
            if (bi.object < 0 || (uint32_t)bi.object >= in->n_objects) return HPRT_E_INVALID;
            HprtInstanceDesc id; id.object = bi.object;
            memcpy(id.instance_to_world, bi.instance_to_world, 64); memcpy(id.world_to_instance, bi.world_to_instance, 64);
            a->topItems.push_back(HprtTopItem{1, (uint32_t)a->instances.size()});
            a->instances.push_back(id);
            nTop += 1;
        } else {
            a->topItems.push_back(HprtTopItem{0, (uint32_t)a->shapes.size()});
            if (int rc = addShape(r)) return rc;
            nTop += primsOf(r);
        }
    }
    if (int rc = BuildAggregate(in->prim_bounds, nTop, *in, &a->top)) return rc;
    uint32_t info[4];
    if (int rc = hprt_bvh_info(a->top.bvh, info, a->bounds)) return rc;
    HprtSceneDesc &d = a->desc; memset(&d, 0, sizeof(d));
    d.nodes = a->top.nodes.data(); d.n_nodes = a->top.nNodes; d.prim_order = a->top.order.data(); d.n_prims = a->top.nPrims;
    d.shapes = a->shapes.data(); d.n_shapes = (uint32_t)a->shapes.size();
    d.materials = in->materials; d.n_materials = in->n_materials;
    d.lights = a->lights.data(); d.n_lights = (uint32_t)a->lights.size(); d.light_strategy = in->light_strategy;
    d.textures = in->textures; d.n_textures = in->n_textures;
    d.objects = a->objects.empty() ? nullptr : a->objects.data(); d.n_objects = (uint32_t)a->objects.size();
    d.instances = a->instances.empty() ? nullptr : a->instances.data(); d.n_instances = (uint32_t)a->instances.size();
    d.top = a->topItems.data(); d.n_top = (uint32_t)a->topItems.size();
    *out = a.release();
    return HPRT_OK;
}

int hprt_bridge_accel_upload(HprtBridgeAccel *a, int device) {
    if (!a) return HPRT_E_INVALID;
    if (a->scene) { hprt_scene_destroy(a->scene); a->scene = nullptr; }
    return hprt_scene_create(&a->desc, device, &a->scene);
}

void hprt_bridge_accel_destroy(HprtBridgeAccel *a) { delete a; }

int hprt_bridge_accel_world_bound(const HprtBridgeAccel *a, float bounds6[6]) {
    if (!a || !bounds6) return HPRT_E_INVALID;
    memcpy(bounds6, a->bounds, 24);
    return HPRT_OK;
}
const HprtBvh *hprt_bridge_accel_bvh(const HprtBridgeAccel *a) { return a ? a->top.bvh : nullptr; }
const HprtSceneDesc *hprt_bridge_accel_desc(const HprtBridgeAccel *a) { return a ? &a->desc : nullptr; }
HprtScene *hprt_bridge_accel_scene(const HprtBridgeAccel *a) { return a ? a->scene : nullptr; }

void hprt_bridge_fill_options(const HprtBridgeFrame *f, HprtRenderOptions *o) {
    memset(o, 0, sizeof(*o));
    o->xres = f->full_resolution[0]; o->yres = f->full_resolution[1];
    memcpy(o->crop, f->crop_window, 16);
    memcpy(o->filter_radius, f->filter_radius, 8);
    o->film_scale = f->film_scale; o->max_sample_luminance = f->max_sample_luminance;
    o->fov = f->fov; o->lens_radius = f->lens_radius; o->focal_distance = f->focal_distance;
    memcpy(o->screen_window, f->screen_window, 16);
    memcpy(o->camera_to_world, f->camera_to_world, 64); memcpy(o->world_to_camera, f->world_to_camera, 64);
    o->spp = f->samples_per_pixel; o->sample_pixel_center = f->sample_at_pixel_center;
    o->max_depth = f->max_depth; o->rr_threshold = f->rr_threshold; o->light_strategy = f->light_strategy;
    o->max_node_prims = f->max_node_prims; o->isect_cost = f->isect_cost; o->trav_cost = f->trav_cost;
}

int hprt_bridge_render(HprtBridgeAccel *a, const HprtRenderDesc *desc, void *pixels, size_t pixel_stride, size_t xyz_offset,
                       size_t weight_offset, HprtRenderStats *stats) {
    if (!a || !a->scene || !desc || !pixels || pixel_stride < 16) return HPRT_E_INVALID;
    int rc = hprt_render(a->scene, desc, nullptr, nullptr, stats);
    if (rc != HPRT_OK) return rc;
    // croppedPixelBounds (core/film.cpp:56-60)
    const HprtRenderOptions &o = desc->opt;
    auto ceilI = [](float v) { int i = (int)v; return (float)i < v ? i + 1 : i; };
    const int x0 = ceilI((float)o.xres * o.crop[0]), x1 = ceilI((float)o.xres * o.crop[1]);
    const int y0 = ceilI((float)o.yres * o.crop[2]), y1 = ceilI((float)o.yres * o.crop[3]);
    const size_t n = (size_t)(x1 - x0) * (size_t)(y1 - y0);
    std::vector<float> xyzw(4 * n);
    if ((rc = hprt_film_read(a->scene, xyzw.data(), n)) != HPRT_OK) return rc;
    char *base = (char *)pixels;
    for (size_t i = 0; i < n; ++i) {      // Film::pixels[i].xyz / .filterWeightSum: what MergeFilmTile would have left (core/film.cpp:118-132)
        memcpy(base + i * pixel_stride + xyz_offset, &xyzw[4 * i], 12);
        memcpy(base + i * pixel_stride + weight_offset, &xyzw[4 * i + 3], 4);
    }
    return HPRT_OK;
}

}  // extern "C"
