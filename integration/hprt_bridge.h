/* hprt_bridge — the one translation unit of the pbrt-side adapters that talks to include/hprt.h.
 *
 * integration/hprt_accel.cpp and integration/hprt_path_integrator.cpp are written against the reference's own headers
 * (core/primitive.h, core/film.h, ...) and cannot be compiled in a tree that lacks the reference's glog / OpenEXR / Ptex
 * submodules.  Everything they do with the C ABI goes through the functions below, on PLAIN structs that mirror the pbrt
 * members they are filled from (named per field), so that this file IS compiled and tested without pbrt
 * (tests/test_integration_bridge.py drives it through ctypes): the adapters are left with member-to-field copies.
 *
 * Ownership: every pointer in the input structs is borrowed for the duration of the call (pbrt keeps owning its meshes);
 * the accel object owns the BVHs it builds and the HprtScene it uploads.  Errors: hprt's codes, message in hprt_last_error().
 */
#ifndef HPRT_BRIDGE_H
#define HPRT_BRIDGE_H
#include "../include/hprt.h"
#ifdef __cplusplus
extern "C" {
#endif

/* TriangleMesh (shapes/triangle.h:54-75) + the Shape flags of its Triangles (core/shape.h:79-80).  Point3f / Normal3f /
 * Vector3f arrays are 3 floats per element, Point2f 2: the unique_ptr<T[]>::get() of the mesh reinterpret_cast to float*. */
typedef struct HprtBridgeMesh {
    int32_t n_triangles, n_vertices;       /* TriangleMesh::nTriangles, nVertices */
    const int32_t *vertex_indices;         /* TriangleMesh::vertexIndices.data(), 3 per triangle */
    const float *p, *n, *s, *uv;           /* TriangleMesh::p / n / s / uv (world space, shapes/triangle.cpp:72-88); n, s, uv may be NULL */
    int32_t reverse_orientation, transform_swaps_handedness;   /* Shape::reverseOrientation, transformSwapsHandedness of its triangles */
    int32_t material;                      /* index into HprtBridgeScene::materials (GeometricPrimitive::material) */
    int32_t first_area_light;              /* -1, or the index in Scene::lights of the DiffuseAreaLight of face 0; faces follow (core/api.cpp:1609-1636) */
} HprtBridgeMesh;

/* Sphere (shapes/sphere.h:50-81) */
typedef struct HprtBridgeSphere {
    float object_to_world[16], world_to_object[16];   /* Shape::ObjectToWorld->GetMatrix().m, WorldToObject->GetMatrix().m, row-major */
    float radius, z_min, z_max, theta_min, theta_max, phi_max;
    int32_t reverse_orientation, transform_swaps_handedness;
    int32_t material;
    int32_t area_light;                    /* -1, or index in Scene::lights */
} HprtBridgeSphere;

/* A run of consecutive entries of a primitive vector (BVHAccel::primitives, or an object definition's list):
 * kind 0: the n_triangles GeometricPrimitives over the Triangles of meshes[index], in face order;
 * kind 1: the GeometricPrimitive over spheres[index];  kind 2: the TransformedPrimitive instances[index] (top level only). */
typedef struct HprtBridgeRun { int32_t kind; int32_t index; } HprtBridgeRun;

/* One object definition (renderOptions->instances[name] before ObjectInstance wraps it, core/api.cpp:1798-1806) */
typedef struct HprtBridgeObject {
    const HprtBridgeRun *runs; uint32_t n_runs;
    const float *prim_bounds;              /* Primitive::WorldBound() of each primitive of the object (its own space), 6 floats each */
} HprtBridgeObject;

/* TransformedPrimitive (core/primitive.h:103-128): PrimitiveToWorld's start transform */
typedef struct HprtBridgeInstance { int32_t object; float instance_to_world[16], world_to_instance[16]; } HprtBridgeInstance;

typedef struct HprtBridgeScene {
    const HprtBridgeMesh *meshes; uint32_t n_meshes;
    const HprtBridgeSphere *spheres; uint32_t n_spheres;
    const HprtBridgeRun *runs; uint32_t n_runs;        /* the primitive vector the aggregate was given, as runs, in its order */
    const float *prim_bounds;                          /* primitives[i]->WorldBound() (pMin, pMax), one per PRIMITIVE in that order */
    const HprtBridgeObject *objects; uint32_t n_objects;
    const HprtBridgeInstance *instances; uint32_t n_instances;
    const HprtMaterialDesc *materials; uint32_t n_materials;   /* include/hprt.h: filled from the Material objects' constant parameters */
    const HprtLightDesc *lights; uint32_t n_lights;            /* Scene::lights order; .shape is filled in by the bridge */
    const HprtTextureDesc *textures; uint32_t n_textures;
    int32_t light_strategy;                                    /* 0 uniform, 1 power, 2 spatial ("lightsamplestrategy") */
    int32_t max_node_prims, isect_cost, trav_cost;             /* CreateBVHAccelerator's parameters (accelerators/bvh.cpp:529-535) */
} HprtBridgeScene;

typedef struct HprtBridgeAccel HprtBridgeAccel;

/* Host part of HprtAccel's constructor: the aggregates (hprt_bvh_build_from_bounds: the reference's sweep-SAH build, node array
 * byte-identical to BVHAccel's) and the HprtSceneDesc.  No GPU needed. */
int hprt_bridge_accel_build(const HprtBridgeScene *in, HprtBridgeAccel **out);
/* Device part: hprt_scene_create on `device` (< 0: current).  The input structs of hprt_bridge_accel_build must still be alive. */
int hprt_bridge_accel_upload(HprtBridgeAccel *a, int device);
void hprt_bridge_accel_destroy(HprtBridgeAccel *a);
/* BVHAccel::WorldBound() (accelerators/bvh.cpp:187-189): pMin, pMax */
int hprt_bridge_accel_world_bound(const HprtBridgeAccel *a, float bounds6[6]);
/* What was built (tests, and adapters that want the pieces): the top-level BVH, the scene description, the device scene */
const HprtBvh *hprt_bridge_accel_bvh(const HprtBridgeAccel *a);
const HprtSceneDesc *hprt_bridge_accel_desc(const HprtBridgeAccel *a);
HprtScene *hprt_bridge_accel_scene(const HprtBridgeAccel *a);

/* The frame, as the pbrt objects hold it */
typedef struct HprtBridgeFrame {
    int32_t full_resolution[2];            /* Film::fullResolution */
    float crop_window[4];                  /* the cropWindow Film was constructed with: x0 x1 y0 y1 (core/film.cpp:56-60) */
    float filter_radius[2];                /* Film::filter->radius */
    float film_scale, max_sample_luminance;/* Film::scale, maxSampleLuminance */
    float camera_to_world[16], world_to_camera[16];   /* Camera::CameraToWorld start transform and its inverse, row-major */
    float fov, lens_radius, focal_distance;/* CameraParams "fov", "lensradius", "focaldistance" (cameras/perspective.cpp:224-271) */
    float screen_window[4];                /* "screenwindow", or the frame-aspect default computed there: x0 x1 y0 y1 */
    int32_t samples_per_pixel;             /* Sampler::samplesPerPixel (HaltonSampler) */
    int32_t sample_at_pixel_center;        /* HaltonSampler::sampleAtPixelCenter */
    int32_t max_depth; float rr_threshold; /* "maxdepth", "rrthreshold" (integrators/path.cpp:209-224) */
    int32_t light_strategy;
    int32_t max_node_prims, isect_cost, trav_cost;
} HprtBridgeFrame;
void hprt_bridge_fill_options(const HprtBridgeFrame *f, HprtRenderOptions *out);

/* HprtPathIntegrator::Render's middle: hprt_render + film read-back into Film::pixels.  `pixels` points at Film::pixels[0];
 * the fork's Film::Pixel is ~224 bytes (its GeneralStats member, core/film.h:85-92), so the layout is given by the caller:
 * pixel i's xyz[3] live at pixels + i * pixel_stride + xyz_offset, filterWeightSum at + weight_offset.  Pixels are written
 * for the cropped pixel bounds, row-major (Film::GetPixel's order). */
int hprt_bridge_render(HprtBridgeAccel *a, const HprtRenderDesc *desc, void *pixels, size_t pixel_stride, size_t xyz_offset,
                       size_t weight_offset, HprtRenderStats *stats);

#ifdef __cplusplus
}
#endif
#endif
