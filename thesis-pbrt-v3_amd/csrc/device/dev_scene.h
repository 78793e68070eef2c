// hprt device side — plain-data view of the scene in HBM, passed to kernels by value.
//
// HBM layout (all arrays 16-byte aligned, allocated once per scene):
//   pairs      DevPair[nPairs]      64 B per INTERIOR node of the reference's LinearBVHNode[]
//                                    (accelerators/bvh.cpp:123-152), holding the bounds of both
//                                    children and references to them, so a traversal step is one
//                                    64-B read (four 16-B requests, one cache-line half) that
//                                    decides both children; leaves have no record of their own.
//                                    Pair 0 is a synthetic parent of the root (PAIR_SINGLE) that
//                                    carries the root's bounds test.  Pairs follow the depth-first
//                                    order of the linear array, so a node and its first child
//                                    stay adjacent.  The LinearBVHNode[] itself stays on the host
//                                    (hprt_bvh_copy hands it out byte-identical).
//   tris       float4[3*nPrims]     48 B/primitive in BVH (ordered) order, vertices
//                                    pre-gathered: {p0,tag} {p1,shape} {p2,aux}.  Replaces the
//                                    reference's primitive -> shape -> mesh -> index -> vertex
//                                    pointer chase (core/primitive.cpp:118-138) by one
//                                    contiguous 48-B read per triangle test.  tag bits: 0-1
//                                    kind (0 triangle, 1 sphere), bit 2 "bogus" (zero-area
//                                    triangle: Triangle::Intersect returns false at
//                                    shapes/triangle.cpp:309-316, IntersectP does not), bit 3
//                                    "last primitive of its leaf" (ends the leaf loop of
//                                    accelerators/bvh.cpp:370-373 without a count).
//                                    For spheres aux = sphere index; for a triangle of an emissive mesh aux =
//                                    1 + the index of its DiffuseAreaLight (0: not a light); kind 2 is an object
//                                    instance (TransformedPrimitive) with aux = instance index.
//                                    The primitives of all aggregates share this array: the top
//                                    level first, then each object definition's.
//   primVtx    uint32[3*nPrims]     global vertex ids of the ordered triangle (shading only)
//   primN      float4[3*nPrims]     the three shading normals of the ordered triangle, gathered like
//                                    its positions: one more contiguous read instead of the index ->
//                                    vertex chase (zeros for meshes without normals)
//   vUV/vS     float[2|3 * nVtx]    remaining per-vertex shading attributes (rare), reached through primVtx
//   shapes, materials, lights, spheres, lightCdf: small tables.
#pragma once
#include "../hprt_math.h"
#include "dev_wide.h"

namespace hprt {

// x/y/z: {lo child0, lo child1, hi child0, hi child1}; child0 = first child (node + 1),
// child1 = secondChildOffset.  ref: >= 0 pair index of an interior child, ~firstPrimitive of a
// leaf child.  meta: split axis (bits 0-1) | PAIR_SINGLE.
struct DevPair { float x[4], y[4], z[4]; int32_t ref0, ref1; uint32_t meta, pad; };
enum : uint32_t { PAIR_SINGLE = 4u };
// The primitive word of a hit record: ordered index (< 2^28) | shading-bin code << 28 (k_bin then needs
// nothing but this word): HIT_PLASTIC triangle of a plastic material, HIT_GENERIC quadric or hit inside an instance
// Shading bins (k_bin -> k_shade variants): matte and plastic and substrate triangles reached directly have variants of their own;
// everything else (quadrics, emitters, hits inside instances, the other materials) goes to the generic variant — the one compiled
// with the MIPMap lookups when the material has an image texture.
enum : uint32_t { BIN_MATTE = 0u, BIN_PLASTIC = 1u, BIN_GENERIC = 2u, BIN_TEXTURED = 3u, BIN_SUBSTRATE = 4u, N_BINS = 5u };
enum : uint32_t { HIT_PRIM_MASK = 0x0fffffffu, HIT_BIN_SHIFT = 28u };      // bits 28-30 of a hit's primitive word: its bin (bit 31 stays the sign: -1 = miss)
__host__ __device__ inline int32_t hit_prim(int32_t word) { return word < 0 ? word : (int32_t)((uint32_t)word & HIT_PRIM_MASK); }

// REF_NONE: nothing left.  REF_EXIT: stack sentinel under an instance's walk — popping it ends the
// instance (TransformedPrimitive::Intersect returns, core/primitive.cpp:77-93)
enum : int32_t { REF_NONE = (int32_t)0x80000000, REF_EXIT = (int32_t)0x80000001 };
enum : int32_t { VOX_EMPTY = -1, VOX_REQUESTED = -2 };

enum : uint32_t { TAG_KIND_MASK = 3u, TAG_SPHERE = 1u, TAG_INSTANCE = 2u, TAG_BOGUS = 4u, TAG_LAST = 8u,
                  TAG_BIN_SHIFT = 4u, TAG_BIN_MASK = 7u << 4,      // bits 4-6: the primitive's shading bin (BIN_*), decided on the host
                  // an instance primitive whose world-to-instance matrix is affine (last row exactly 0 0 0 1) carries the matrix's 3x3
                  // part in the x, y, z words of its three record words; translation and entry sit in DevScene::topEntry[primitive]
                  TAG_INST_INLINE = 128u };
// (tag & TAG_BIN_MASK) << 24 is the bin field of the hit word: k_bin needs nothing but that word
enum : uint32_t { SHAPE_FLIP = 1u, SHAPE_HAS_N = 2u, SHAPE_HAS_UV = 4u, SHAPE_HAS_S = 8u, SHAPE_REVERSE = 16u };

struct DevShape { int32_t material, areaLight; uint32_t flags; int32_t sphere; };
struct DevMaterial { int32_t type; float Kd[3]; float Ks[3]; float alpha; int32_t KdTex, KsTex; float orenA, orenB; int32_t oren; float alphaY;
                     float Kr[3], Kt[3], opacity[3], eta; int32_t opTex; int32_t roughGlass; };      // roughGlass: type 5 with Kr[0], Kr[1] = the Trowbridge-Reitz alphas of its two microfacet lobes      // (type 6, uber: Kr, Kt, opacity, eta; alpha / alphaY from u / vroughness)      // type 2: mirror, Kr in Ks; 3: substrate (alpha, alphaY = TR alphas of u / vroughness); 4: metal (Kd = eta, Ks = k); 5: smooth glass (Kd = Kt, Ks = Kr, alpha = eta); oren: OrenNayar's A, B (core/reflection.h:414-420) evaluated on the host   // alpha: RoughnessToAlpha applied on the host; *Tex: image texture or -1
// ImageTexture + MIPMap (textures/imagemap.h, core/mipmap.h): levels are consecutive in mipLevels, texels hold 3 floats each
struct DevMipLevel { uint32_t offset; int32_t w, h; };
struct DevTexture { uint32_t firstLevel, nLevels; int32_t trilinear, wrap; float maxAniso, su, sv, du, dv; };
// type: 0 point, 1 distant, 2 diffuse area light on a sphere, 3 diffuse area light on ONE triangle (prim = its ordered index;
// every triangle of an emissive mesh is a light of its own, core/api.cpp:1609-1636)
struct DevLight { int32_t type; float pos[3]; float I[3]; int32_t shape; int32_t twoSided; int32_t sphere; uint32_t shapeFlags; int32_t prim; };
struct DevSphere { mat4 o2w, w2o; float radius, zMin, zMax, thetaMin, thetaMax, phiMax; };
// InfiniteAreaLight (lights/infinite.cpp): DevLight type 4, DevLight::shape indexes this table.  Its Distribution2D (core/sampling.h:
// 128-151) sits in DevScene::envData at `off`: condFunc[nv][nu], condCdf[nv][nu + 1], condFuncInt[nv] (= the marginal's func),
// margCdf[nv + 1]; margFuncInt here.
struct DevEnvLight { mat4 l2w, w2l; int32_t tex; int32_t nu, nv; uint32_t off; float margFuncInt; float pad[3]; };
// ObjectInstance: the wrapped aggregate's entry (pair index, or ~primitive when the object holds a
// single primitive: no aggregate, no bounds test) and the static instance transform.  pad[0]: the same entry
// into the WIDE records (k_walk4): first wide record of the object's tree, or that one primitive as a leaf reference
// in the "box already passed" state
struct DevInstance { mat4 i2w, w2i; int32_t root; uint32_t identity; uint32_t pad[2]; };

struct DevScene {
    const DevPair *pairs; uint32_t nPairs;
    // The leaf-exact walk of plain renders (dev_wide.h; null when the scene keeps the binary walk): four-wide records over the same
    // leaves, and per ordered primitive the exact box of its leaf {lo.xyz, hi.x} {hi.yz, -, -} for leaves that are not one triangle
    const DevWide *wide; uint32_t nWide;
    const float4 *leafBox;
    const float4 *tris; uint32_t nPrims;
    const uint32_t *primVtx;
    const float4 *primN;                                         // shading normals pre-gathered like the positions: float4[3] per ordered primitive
    const float *vUV, *vS;
    const DevShape *shapes; uint32_t nShapes;
    const DevMaterial *materials;
    const DevLight *lights; uint32_t nLights;
    const DevTexture *textures; const DevMipLevel *mipLevels; const float *texels; const float *weightLut;   // image textures; weightLut: MIPMap::weightLut[128]
    const DevSphere *spheres; uint32_t nSpheres;
    const DevEnvLight *envLights; const float *envData; uint32_t nEnvLights;      // infinite lights (escaped rays pick up their radiance)
    const DevInstance *instances; uint32_t nInstances;
    // per top-level primitive (instanced scenes; null when there are none): {m03, m13, m23 of the instance's world-to-instance matrix, its
    // entry (DevInstance::root)}, fetched together with the primitive record so that entering an instance costs one memory round trip, not two
    const float4 *topEntry; uint32_t nTopPrims;
    const float4 *topEntryWide;                                  // the same with the instance's entry into the WIDE records (k_walk4); DevInstance::pad[0] holds it too
    const float *lightFunc, *lightCdf; float lightFuncInt;      // Distribution1D (core/sampling.h:55-109): uniform or power strategy
    // SpatialLightDistribution (core/lightdistrib.cpp:77-300), computed for every voxel at scene creation: per voxel v the
    // Distribution1D's func[nLights] at voxFunc + v * nLights, cdf[nLights + 1] at voxCdf + v * (nLights + 1), funcInt at voxFuncInt[v]
    int32_t spatial; int32_t voxN[3]; float wbMin[3], wbMax[3];
    const float *voxFunc, *voxCdf, *voxFuncInt;
    // On-demand mode (the table of EVERY voxel would be too large: many lights): voxSlot[v] = row of voxel v in the tables above, or
    // VOX_EMPTY / VOX_REQUESTED; a vertex that falls into a voxel without a row asks for it (voxRequest / voxRequestCount) and is
    // shaded again once the host has had the rows computed — the reference's lazy fill (core/lightdistrib.cpp:149-229), per batch
    // instead of per thread.  voxSlot == nullptr: every voxel has its row (row = voxel).
    int32_t *voxSlot; uint32_t *voxRequest; uint32_t *voxRequestCount;
    float worldRadius;                                           // DistantLight::Preprocess
    uint2 *deepStack;                                            // traversal stack entries beyond the LDS ones: [entry][grid thread]
    // Halton tables
    const uint16_t *perms; const int32_t *primes; const int32_t *primeSums; const uint64_t *primeMagic;
};

// counters accumulated by the traversal kernels when counting is requested
struct DevCounters {
    unsigned long long nodesFetched, nodesEntered, triTests, sphereTests;         // closest hit
    unsigned long long nodesFetchedP, nodesEnteredP, triTestsP, sphereTestsP;     // any hit
};

}  // namespace hprt
