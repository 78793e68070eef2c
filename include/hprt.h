/* hprt — C ABI of the MI355X-native wavefront path-tracing core.
 *
 * Drop-in boundary for ONE hot path of the pbrt-v3 thesis fork
 * (jhoobergs/Thesis-pbrt-v3):
 *   SamplerIntegrator::Render  ->  PathIntegrator::Li  ->
 *   BVHAccel::Intersect/IntersectP  ->  Triangle::Intersect/IntersectP.
 * Plain pointers and sizes only; no C++ or torch types cross this header.
 * Every entry point names the reference interface it stands in for
 * (paths relative to the reference's src/).  All functions return 0 on success
 * or a negative HPRT_E_* code; hprt_last_error() returns the message of the last
 * failure on the calling thread.  No exception crosses the boundary.
 *
 * Device entry points (hprt_scene_*, hprt_intersect, hprt_occluded, hprt_render,
 * hprt_film_*) require a gfx950 GPU and fail with HPRT_E_NO_DEVICE otherwise:
 * there is no CPU fallback behind this ABI.
 *
 * Concurrency.  An HprtScene allows ONE call in flight at a time: its work counter,
 * ray / hit staging, traversal-stack area and render workspace belong to the scene,
 * not to a call.  The library enforces it — calls from several host threads take the
 * scene's mutex, and a call on one HIP stream first waits (on the device) for an
 * earlier asynchronous *_device call on another — so concurrent use is SAFE but
 * serialised: where pbrt calls BVHAccel::Intersect from every worker thread
 * (core/parallel.cpp:247-299), a host gains nothing by doing the same here; it should
 * batch.  HprtModel / HprtBvh objects are immutable after creation and may be read
 * from any number of threads.  Different HprtScene objects are independent.
 */
#ifndef HPRT_H
#define HPRT_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define HPRT_OK 0
#define HPRT_E_INVALID (-1)     /* bad argument / malformed description */
#define HPRT_E_IO (-2)          /* file could not be read or written     */
#define HPRT_E_PARSE (-3)       /* scene description rejected            */
#define HPRT_E_NO_DEVICE (-4)   /* no usable HIP device                  */
#define HPRT_E_DEVICE (-5)      /* HIP runtime error                     */
#define HPRT_E_UNSUPPORTED (-6) /* feature outside the hot-path scope    */

const char *hprt_last_error(void);
/* Compile-time identity of the library: "hprt <ver> gfx950 ..." */
const char *hprt_version(void);

/* ------------------------------------------------------------------------ */
/* Host front-end: the parsed scene ("model").                               */
/* Stands in for pbrtParseFile + the pbrt* API state machine                 */
/* (core/parser.cpp:786-1092, core/api.cpp:1103-1874) up to, but excluding,  */
/* MakeScene()/Render().                                                     */
/* ------------------------------------------------------------------------ */
typedef struct HprtModel HprtModel;

typedef struct HprtRenderOptions {
    int32_t xres, yres;            /* Film "xresolution"/"yresolution" (core/film.cpp:322-323) */
    float crop[4];                 /* cropwindow x0 x1 y0 y1 (core/film.cpp:326-341)           */
    float filter_radius[2];        /* box filter half-widths (filters/box.cpp:43-47)            */
    float film_scale, max_sample_luminance;
    float fov, lens_radius, focal_distance;      /* cameras/perspective.cpp:224-271 */
    float screen_window[4];        /* x0 x1 y0 y1 */
    float camera_to_world[16], world_to_camera[16]; /* row-major Transform::m / mInv */
    int32_t spp;                   /* Sampler "pixelsamples" (samplers/halton.cpp:135)          */
    int32_t sample_pixel_center;
    int32_t max_depth;             /* Integrator "maxdepth" (integrators/path.cpp:209)          */
    float rr_threshold;            /* integrators/path.cpp:224                                   */
    int32_t light_strategy;        /* 0 uniform, 1 power, 2 spatial (core/lightdistrib.cpp:47-66); all three built */
    int32_t max_node_prims, isect_cost, trav_cost; /* accelerators/bvh.cpp:529-535 */
} HprtRenderOptions;

/* Parse a .pbrt file.  `subst` holds n_subst {key,value} string pairs that
 * replace the fork's template tokens ($acc -> "bvh", $accnr -> 0, ... as
 * scripts/render_simple.sh:23-29 does with sed) and, for keys that end in '/',
 * path prefixes of Include/plymesh file names. */
int hprt_model_parse(const char *pbrt_path, const char *const *subst, int n_subst, HprtModel **out);
/* Image textures of a parsed model (built MIPMaps): count[0] = textures; info = {levels, trilinear, wrap, width
 * and height of level 0}; hprt_model_texture_level copies 3*w*h floats of one level (rgb may be NULL to query w,h). */
int hprt_model_texture_info(const HprtModel *m, uint32_t texture, int32_t info[5], float *max_anisotropy);
int hprt_model_texture_level(const HprtModel *m, uint32_t texture, uint32_t level, int32_t wh[2], float *rgb);
/* Baked scene container (post-parse, world-space; DESIGN.md "Baked scene"). */
int hprt_model_load(const char *baked_path, HprtModel **out);
int hprt_model_save(const HprtModel *m, const char *baked_path);
/* The same with every image texture stored as the image it was read from (8-bit texels as the file held them + the
 * conversion parameters of ImageTexture::GetTexture, textures/imagemap.cpp:52-97) instead of its finished float pyramid:
 * 3 bytes per source texel instead of 16 per power-of-two texel.  hprt_model_load rebuilds the pyramid with the MIPMap
 * constructor that built it at parse time (core/mipmap.h:113-201), so both forms load to the same floats. */
int hprt_model_save_compact(const HprtModel *m, const char *baked_path);
void hprt_model_destroy(HprtModel *m);
int hprt_model_get_options(const HprtModel *m, HprtRenderOptions *out);
int hprt_model_set_options(HprtModel *m, const HprtRenderOptions *in);
/* counts[0..6] = shapes, primitives, triangles, spheres, materials, lights, image textures */
int hprt_model_counts(const HprtModel *m, uint64_t counts[7]);
/* Front-end warnings (out-of-scope features that were substituted), '\n' separated. */
const char *hprt_model_warnings(const HprtModel *m);

/* ------------------------------------------------------------------------ */
/* Accelerator build.  Stands in for BVHAccel::BVHAccel + iterativeBuild +   */
/* flattenBVHTree (accelerators/bvh.cpp:155-185, 196-333, 335-350) and       */
/* CreateBVHAccelerator (:529-535).  Host side; the node array is            */
/* byte-identical to the reference's LinearBVHNode[] (:123-152).             */
/* ------------------------------------------------------------------------ */
typedef struct HprtBvh HprtBvh;
int hprt_bvh_build(const HprtModel *m, HprtBvh **out);
/* The same from the primitives' world bounds (Primitive::WorldBound(), core/primitive.h:55):
 * bmin/bmax hold 3 floats per primitive in creation order. */
int hprt_bvh_build_from_bounds(size_t n_prims, const float *bmin, const float *bmax, int max_node_prims, int isect_cost,
                               int trav_cost, HprtBvh **out);
void hprt_bvh_destroy(HprtBvh *b);
/* info[0..3] = nodes, primitives, leaves, max depth; bounds6 = root pMin,pMax
 * (BVHAccel::WorldBound, accelerators/bvh.cpp:187-189) */
int hprt_bvh_info(const HprtBvh *b, uint32_t info[4], float bounds6[6]);
/* nodes32: n_nodes*32 bytes; prim_order: n_prims uint32 (ordered -> creation number) */
int hprt_bvh_copy(const HprtBvh *b, void *nodes32, uint32_t *prim_order);
/* The same for the aggregate of object definition `object` (core/api.cpp:1798-1806). */
int hprt_bvh_object_info(const HprtBvh *b, uint32_t object, uint32_t info[4], float bounds6[6]);
int hprt_bvh_object_copy(const HprtBvh *b, uint32_t object, void *nodes32, uint32_t *prim_order);

/* ------------------------------------------------------------------------ */
/* Device scene.  Upload step that follows the BVH build: stands in for the  */
/* `primitives`/`nodes` members BVHAccel keeps (accelerators/bvh.h:69-79) and */
/* the Scene object (core/scene.h:50-80).  The library copies everything to   */
/* HBM and owns that memory until hprt_scene_destroy.                         */
/* ------------------------------------------------------------------------ */
typedef struct HprtScene HprtScene;

typedef struct HprtShapeDesc {       /* one Shape directive (core/api.cpp:1561-1651) */
    int32_t kind;                    /* 0 triangle mesh, 1 sphere */
    int32_t material;                /* index into materials */
    int32_t area_light;              /* index into lights or -1 (GeometricPrimitive::areaLight).  A mesh is n_tris Triangle shapes, each with
                                      * a DiffuseAreaLight of its own (core/api.cpp:1609-1636): lights area_light .. area_light + n_tris - 1,
                                      * in face order, each with .shape = this shape */
    int32_t reverse_orientation, transform_swaps_handedness; /* core/shape.h:79-80 */
    /* mesh: world-space arrays as TriangleMesh holds them (shapes/triangle.cpp:54-92) */
    uint32_t n_tris, n_verts;
    const int32_t *indices;          /* 3*n_tris */
    const float *P;                  /* 3*n_verts */
    const float *N;                  /* 3*n_verts or NULL */
    const float *UV;                 /* 2*n_verts or NULL */
    const float *S;                  /* 3*n_verts or NULL */
    /* sphere (shapes/sphere.h:50-59) */
    float object_to_world[16], world_to_object[16];
    float radius, z_min, z_max, theta_min, theta_max, phi_max;
} HprtShapeDesc;

typedef struct HprtMaterialDesc {    /* materials/matte.cpp:64-72, materials/plastic.cpp:72-84, materials/mirror.cpp:58-64 */
    int32_t type;                    /* 0 matte (sigma != 0: OrenNayar), 1 plastic, 2 mirror (Kr in Ks), 3 substrate (roughness = uroughness,
                                      * sigma = vroughness; materials/substrate.cpp), 4 metal (Kd = eta, Ks = k, roughness / sigma likewise;
                                      * materials/metal.cpp), 5 glass (Kd = Kt, Ks = Kr, roughness = eta; sigma = uroughness and Kr[0] = vroughness — both 0: the smooth FresnelSpecular
                                      * lobe, else MicrofacetReflection + MicrofacetTransmission; materials/glass.cpp:61-93), 6 uber (below) */
    float Kd[3], sigma, Ks[3], roughness;
    int32_t remap_roughness;
    int32_t kd_texture, ks_texture;  /* index into HprtSceneDesc::textures when Kd / Ks is an image texture, else -1 */
    /* type 6, UberMaterial (materials/uber.cpp): Kd, Ks as named, roughness = uroughness, sigma = vroughness, and the lobes only it
     * has: Kr (specular reflection), Kt (specular transmission), opacity (1 - opacity passes straight through), eta */
    float Kr[3], Kt[3], opacity[3], eta;
    int32_t opacity_texture;         /* uber: "opacity" as an image texture (materials/uber.cpp:53, scenes/livingroom:30), else -1 */
} HprtMaterialDesc;

/* ImageTexture<RGBSpectrum, Spectrum> (textures/imagemap.h:71-122) with its built MIPMap (core/mipmap.h):
 * level 0 is the power-of-two image after ImageTexture::GetTexture's conversion (scale, inverse gamma, y flip:
 * (0,0) is the lower left texel), level k the 2x2 box filter of level k-1; rgb holds 3*w*h floats per level. */
typedef struct HprtTextureLevel { int32_t w, h; const float *rgb; } HprtTextureLevel;
typedef struct HprtTextureDesc {
    const HprtTextureLevel *levels; uint32_t n_levels;
    int32_t trilinear;               /* "trilinear" (default false: EWA, core/mipmap.h:262-290) */
    float max_anisotropy;            /* "maxanisotropy", 8 */
    int32_t wrap;                    /* 0 repeat, 1 black, 2 clamp */
    float su, sv, du, dv;            /* UVMapping2D (core/texture.cpp:93-99) */
    const float *weight_lut;         /* MIPMap::weightLut, 128 floats */
} HprtTextureDesc;

typedef struct HprtLightDesc {       /* lights/point.cpp, lights/distant.cpp, lights/diffuse.cpp, lights/infinite.cpp */
    int32_t type;                    /* 0 point, 1 distant, 2 diffuse area, 3 infinite */
    float pos[3];                    /* point: pLight (world); distant: wLight (world, normalised) */
    float I[3];                      /* I / L / Lemit */
    int32_t shape;                   /* area light: shape index */
    int32_t two_sided;
    /* infinite: the radiance map is textures[texture] — its texels as InfiniteAreaLight holds them (ReadImage order, not flipped,
     * already multiplied by L * scale; 1x1 for a constant light) — and the light <-> world transform, row-major */
    int32_t texture;
    float light_to_world[16], world_to_light[16];
} HprtLightDesc;

/* Object instancing (pbrtObjectBegin/End/Instance, core/api.cpp:1752-1820; TransformedPrimitive,
 * core/primitive.cpp:70-102).  An object definition owns a contiguous range of `shapes` and the
 * aggregate ObjectInstance builds over their primitives (core/api.cpp:1798-1806; with a single
 * primitive the reference wraps that primitive directly and the one-leaf tree is only used for
 * its bounds).  An instance is one primitive of the top-level aggregate. */
typedef struct HprtObjectDesc {
    uint32_t first_shape, n_shapes;
    const void *nodes; uint32_t n_nodes;            /* LinearBVHNode layout, primitives numbered within the object */
    const uint32_t *prim_order; uint32_t n_prims;
} HprtObjectDesc;
typedef struct HprtInstanceDesc {
    int32_t object;
    float instance_to_world[16], world_to_instance[16];   /* row-major Transform::m / mInv at ObjectInstance */
} HprtInstanceDesc;
typedef struct HprtTopItem {         /* renderOptions->primitives in creation order */
    int32_t kind;                    /* 0: all primitives of shapes[index]; 1: instances[index] */
    uint32_t index;
} HprtTopItem;

typedef struct HprtSceneDesc {
    const void *nodes;               /* n_nodes * 32 B, LinearBVHNode layout: the top-level aggregate */
    uint32_t n_nodes;
    const uint32_t *prim_order;      /* n_prims: ordered position -> creation-order primitive number */
    uint32_t n_prims;
    const HprtShapeDesc *shapes; uint32_t n_shapes;   /* creation order */
    const HprtMaterialDesc *materials; uint32_t n_materials;
    const HprtLightDesc *lights; uint32_t n_lights;
    int32_t light_strategy;
    const HprtTextureDesc *textures; uint32_t n_textures;
    /* instancing; all NULL / 0 without it.  top == NULL: every shape, in order, is top-level */
    const HprtObjectDesc *objects; uint32_t n_objects;
    const HprtInstanceDesc *instances; uint32_t n_instances;
    const HprtTopItem *top; uint32_t n_top;
} HprtSceneDesc;

/* device < 0 selects the current HIP device. */
int hprt_scene_create(const HprtSceneDesc *desc, int device, HprtScene **out);
/* Convenience: the same from a parsed model and its BVH. */
int hprt_scene_create_from_model(const HprtModel *m, const HprtBvh *b, int device, HprtScene **out);
void hprt_scene_destroy(HprtScene *s);

/* ------------------------------------------------------------------------ */
/* Batched Aggregate interface.  Stand in for                                */
/*   bool BVHAccel::Intersect(const Ray&, SurfaceInteraction*) const         */
/*        (accelerators/bvh.cpp:354-396, core/primitive.h:57-61)             */
/*   bool BVHAccel::IntersectP(const Ray&) const (accelerators/bvh.cpp:398-437) */
/* over n rays held in host memory (SoA-of-arrays: o and d are 3*n floats,   */
/* xyz interleaved per ray).  Closest hit writes the shrunken tMax (unchanged */
/* on a miss), the ORDERED primitive index (-1 on a miss) and b0,b1,b2        */
/* (triangles; 0 for spheres).  counters (may be NULL) receives               */
/* [0] BVH nodes fetched (traversal-loop iterations), [1] nodes entered (the   */
/* reference's "BVH node traversals" counter), [2] triangle tests, [3] sphere  */
/* tests — the figures SURVEY.md §8(d)'s byte model is built from.            */
/* These calls are BATCH interfaces: n = 1 meets the single-ray contract of    */
/* Aggregate::Intersect(const Ray&, SurfaceInteraction*) but costs a host-to-  */
/* device copy, a kernel launch and a copy back (~tens of microseconds) per    */
/* ray — a contract shim for tests, not a usable rendering path.  A host that  */
/* wants images calls hprt_render; one that wants rays answered hands over     */
/* thousands to millions at a time.                                            */
/* ------------------------------------------------------------------------ */
int hprt_intersect(HprtScene *s, size_t n, const float *o, const float *d, const float *tmax, float *t_out,
                   int32_t *prim_out, float *bary_out, uint64_t counters[4]);
int hprt_occluded(HprtScene *s, size_t n, const float *o, const float *d, const float *tmax, uint8_t *occluded_out,
                  uint64_t counters[4]);
/* With object instances (TransformedPrimitive::Intersect, core/primitive.cpp:77-93) the ordered
 * primitive index numbers the primitives of all aggregates — top level first, then object 0, 1, ... —
 * and inst_out (may be NULL) receives the instance the hit went through (index into
 * HprtSceneDesc::instances), -1 for none.  hprt_intersect is this call without inst_out. */
int hprt_intersect_instanced(HprtScene *s, size_t n, const float *o, const float *d, const float *tmax, float *t_out,
                             int32_t *prim_out, int32_t *inst_out, float *bary_out, uint64_t counters[4]);
/* Same with rays/hits already resident in HBM (device pointers, SoA planes:
 * ox,oy,oz,dx,dy,dz,tmax each n floats).  `stream` is a hipStream_t or NULL.
 * Timed by bench.py's kernel microbenchmarks. */
int hprt_intersect_device(HprtScene *s, size_t n, const float *d_rays7, float *d_t, int32_t *d_prim, float *d_bary3,
                          void *stream);
int hprt_occluded_device(HprtScene *s, size_t n, const float *d_rays7, uint8_t *d_occ, void *stream);

/* ------------------------------------------------------------------------ */
/* Integrator.  Stands in for SamplerIntegrator::Render(const Scene&)        */
/* (core/integrator.cpp:230-360) with PathIntegrator::Li                     */
/* (integrators/path.cpp:64-204) as the radiance estimator, the Halton        */
/* sampler (samplers/halton.cpp) and the box-filtered Film                    */
/* (core/film.h:130-170, core/film.cpp:118-132).                              */
/* ------------------------------------------------------------------------ */
typedef struct HprtRenderDesc {
    HprtRenderOptions opt;
    /* 16x16 image tiles [tile_begin, tile_end) of the row-major tile grid
     * (core/integrator.cpp:237-244) are rendered; tile_stride > 1 takes every
     * tile_stride-th tile starting at tile_begin (round-robin sharding across
     * GPUs).  tile_end <= 0 means "all tiles". */
    int32_t tile_begin, tile_end, tile_stride;
    int32_t spp_chunk;             /* samples per pixel per wavefront batch; <= 0: automatic */
    int32_t flags;                 /* HPRT_RENDER_* */
} HprtRenderDesc;
#define HPRT_RENDER_COUNT_WORK 1   /* collect node/triangle counters (slower) */
#define HPRT_RENDER_PIXEL_STATS 2  /* also keep them per pixel: the fork's GeneralStats heat-map data (implies COUNT_WORK) */
/* EstimateDirect's BSDF-sampled ray (core/integrator.cpp:176-190) is only traced to learn whether its closest hit is the
 * emitter.  A plain render does not trace it when a cheap exact test proves that it cannot reach the emitter's sphere
 * (it would add exactly zero), nor the segment that leaves a path's last vertex (the reference intersects it and stops,
 * integrators/path.cpp:97-110): same film, fewer rays.  A counting render traces every ray the reference traces, so that
 * its counters are the reference's — unless COUNT_TRACED asks it to count what a plain render traces.  TRACE_ALL makes a
 * plain render trace the reference's full ray set too. */
#define HPRT_RENDER_COUNT_TRACED 4
#define HPRT_RENDER_TRACE_ALL 8
/* Tile-sharded renders whose films hprt_film_gather will merge: box-filter contributions that cross a tile border
 * (FilmTile pixels outside the tile's own 16x16 block, core/film.cpp:98-103) are not merged into the film but kept as
 * HprtFilmRecords, so that the gather can merge the records of ALL ranks into each pixel in source-tile order. */
#define HPRT_RENDER_EXPORT_FOREIGN 16

typedef struct HprtRenderStats {
    uint64_t camera_rays;          /* nCameraRays, core/integrator.cpp:48,293 */
    uint64_t rays;                 /* "Regular ray intersection tests", core/scene.cpp:40,47 */
    uint64_t shadow_rays;          /* "Shadow ray intersection tests",  core/scene.cpp:42,53 */
    uint64_t nodes_fetched, nodes_fetched_p;   /* traversal-loop iterations (closest / any hit) */
    uint64_t nodes_entered, nodes_entered_p;   /* nbNodeTraversals / nbNodeTraversalsP, bvh.cpp:48-49 */
    uint64_t tri_tests, tri_tests_p;           /* nTests / nTestsP, shapes/triangle.cpp:43-44 */
    uint64_t sphere_tests, sphere_tests_p;
    double render_seconds;         /* the reference's Timings/Rendertime span: tile loop only */
    double extend_seconds, occluded_seconds;  /* HIP-event time inside the traversal kernels */
    uint64_t extend_launches, occluded_launches;
    uint64_t extend_rays, occluded_rays;
} HprtRenderStats;

/* Renders into the scene's film.  d_film_xyzw, if not NULL, is a caller-owned
 * DEVICE buffer of 4*W*H floats (W,H = cropped pixel bounds) that receives the
 * merged film state (xyz, filterWeightSum) of the rendered tiles and zeros
 * elsewhere — what Film::MergeFilmTile leaves in Film::pixels.  Summing such
 * buffers over GPUs (RCCL reduce) reproduces the single-GPU film exactly. */
int hprt_render(HprtScene *s, const HprtRenderDesc *desc, float *d_film_xyzw, void *stream, HprtRenderStats *stats);
/* Optional: allocate everything the coming hprt_render(s, desc, ...) needs in HBM now (the wavefront workspace is
 * ~393 B per path of a batch: 105 GB for the 256 M-path batches of a 700x700, 1,024 spp frame), so that a host that
 * renders once (pbrt does: Integrator::Render, core/api.cpp:1851) pays the allocation at scene load — next to
 * BVHAccel's own node allocation (accelerators/bvh.cpp:181) — not inside Render().  A later render with a
 * description that needs more simply grows the buffers. */
int hprt_scene_reserve(HprtScene *s, const HprtRenderDesc *desc);
/* Pixel::stats (core/film.h:91; GeneralStats, core/geometry.h:1078-1173) of the last hprt_render that had
 * HPRT_RENDER_PIXEL_STATS set: per film pixel, row-major over the cropped pixel bounds, 7 values —
 * rays (= samples), primitiveIntersections, primitiveIntersectionsP, leafNodeTraversals,
 * leafNodeTraversalsP, bvhTreeNodeTraversals, bvhTreeNodeTraversalsP — every ray of a pixel's samples
 * adds its counters once (core/integrator.cpp:327-328, integrators/path.cpp:92-200, core/light.cpp:62).
 * Pixels of tiles that were not rendered hold zeros, so per-rank results add up like the film. */
int hprt_pixel_stats_read(HprtScene *s, uint64_t *out7, size_t n_pixels);
/* Film::WriteGeneralStats (core/film.cpp:170-264): writes <prefix>-primitiveIntersections.txt,
 * -primitiveIntersectionsP.txt, -leafNodeTraversals.txt, -leafNodeTraversalsP.txt (one row of the
 * image per line, values separated by blanks) and the all-zero kd-tree / BSP matrices the fork
 * writes for a BVH render.  Its -renderTime.txt (wall-clock per pixel) has no counterpart here. */
int hprt_write_pixel_stats(const char *prefix, const uint64_t *stats7, int width, int height);
/* Film::WriteImage arithmetic (core/film.cpp:266-303) on a host copy of a film
 * state: rgb_out = 3*W*H floats, top row first. */
int hprt_film_resolve(const float *xyzw, size_t n_pixels, float film_scale, float *rgb_out);
/* Film state of the last hprt_render on this scene, copied to the host. */
int hprt_film_read(HprtScene *s, float *xyzw_out, size_t n_pixels);
/* imageio.cpp:437+ : PFM writer (bottom row first, little endian). */
int hprt_write_pfm(const char *path, const float *rgb, int width, int height);

/* ------------------------------------------------------------------------ */
/* Multi-GPU film gather.  Stands in for Film::MergeFilmTile                  */
/* (core/film.cpp:118-132) across GPUs: the reference merges every worker's   */
/* FilmTile into Film::pixels under a mutex; here rank r of n renders tiles   */
/* r, r+n, ... (HprtRenderDesc::tile_begin / tile_stride) with                */
/* HPRT_RENDER_EXPORT_FOREIGN and ONE RCCL step over xGMI lands the frame on  */
/* the root: ncclReduce(sum) of the per-rank films (disjoint addends: exact)  */
/* then a grouped ncclSend/ncclRecv of the few cross-tile records, which the  */
/* root adds per pixel in ascending source-tile order — the order of the      */
/* single-GPU film, so the n-GPU film equals it bit for bit.                  */
/* ------------------------------------------------------------------------ */
typedef struct HprtComm HprtComm;
#define HPRT_COMM_ID_BYTES 128
/* One process per GPU: rank 0 draws an id (ncclGetUniqueId), the host program hands the 128 bytes to every
 * rank by whatever means it has, and every rank creates its communicator (ncclCommInitRank) on `device`
 * (< 0: the current HIP device).  RCCL refuses two ranks on one device. */
int hprt_comm_unique_id(uint8_t id[HPRT_COMM_ID_BYTES]);
int hprt_comm_create(const uint8_t id[HPRT_COMM_ID_BYTES], int rank, int n_ranks, int device, HprtComm **out);
/* rank, size and device as the communicator reports them (ncclCommUserRank / Count / CuDevice); any may be NULL */
int hprt_comm_info(const HprtComm *c, int *rank, int *n_ranks, int *device);
void hprt_comm_destroy(HprtComm *c);
/* Collective over the communicator, after each rank's hprt_render(..., HPRT_RENDER_EXPORT_FOREIGN).  d_film_xyzw is the
 * DEVICE buffer that render wrote (NULL: whichever buffer it wrote — the caller's or the scene's own; a different
 * pointer is refused); on return the root's buffer holds the merged frame (what Film::pixels holds before WriteImage),
 * other ranks' buffers are unspecified.  Blocks until `stream` is done.
 * Errors are collective-safe: a rank whose own arguments or state are unusable still takes part in the first (count)
 * exchange and reports the failure through it, so EVERY rank returns an error and none waits for a peer that left;
 * an error of the reduce or inside the send / recv group is returned only after the group is closed. */
int hprt_film_gather(HprtComm *c, HprtScene *s, float *d_film_xyzw, size_t n_pixels, int root, void *stream);
/* The same for ONE process that drives n GPUs with one HprtScene each (how an adapter inside pbrt would: the proposal
 * of SURVEY.md §8(b)); communicators come from ncclCommInitAll on first use.  d_films may be NULL (every scene's own film). */
int hprt_film_gather_local(HprtScene *const *per_gpu, float *const *d_films, int n, size_t n_pixels, int root);
/* Destroys the communicators hprt_film_gather_local created (ncclCommDestroy) and frees its staging buffers.  Call it
 * before the process ends while the HIP runtime is still alive; the library never does so from a static destructor. */
void hprt_film_gather_local_shutdown(void);
/* The transport-free halves, for hosts that move the data themselves (the gloo rehearsals in tests/ do): the records of
 * the last HPRT_RENDER_EXPORT_FOREIGN render (out == NULL: count only), and the ordered merge of any ranks' records into
 * a HOST copy of the summed films (sorts `records` by destination pixel and source tile, then adds; no GPU involved). */
typedef struct HprtFilmRecord {
    uint32_t dest_pixel;             /* row-major index into the cropped film */
    uint32_t src_tile;               /* tile (core/integrator.cpp:237-244 grid) whose samples these are */
    float xyz[3], weight;            /* that FilmTile pixel's contribSum as XYZ and its filterWeightSum */
} HprtFilmRecord;
int hprt_film_records_read(HprtScene *s, HprtFilmRecord *out, size_t capacity, size_t *n_records);
int hprt_film_records_merge(float *xyzw, size_t n_pixels, HprtFilmRecord *records, size_t n_records);

/* Radiance of individual camera samples (pixel x, y, sample index), after the
 * NaN/negative/inf guards of core/integrator.cpp:300-321; L_out = 3*n floats.
 * Test hook for per-sample parity. */
int hprt_sample_radiance(HprtScene *s, const HprtRenderOptions *opt, size_t n, const int32_t *px, const int32_t *py,
                         const int64_t *sample, float *L_out);

/* Halton sampler tables (host): ComputeRadicalInversePermutations
 * (core/lowdiscrepancy.cpp:2490-2504) with the default-seeded PCG32. */
int hprt_halton_permutations(uint16_t *out, size_t max_entries, size_t *n_entries);

#ifdef __cplusplus
}
#endif
#endif /* HPRT_H */
