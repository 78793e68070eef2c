// hprt host side — the scene as pbrt's MakeScene()/MakeIntegrator() would hold it
// after parsing (reference: core/api.cpp:1883-1946): render options, materials,
// shapes in creation order with world-space mesh data, lights.  Also the baked
// container ("HPRTSCN1") that lets a parsed scene travel without its .pbrt
// sources (layout: DESIGN.md §Baked scene).
#pragma once
#include <map>
#include <string>
#include <vector>
#include "hprt_math.h"

namespace hprt {

enum MaterialType { kMatte = 0, kPlastic = 1, kMirror = 2, kSubstrate = 3, kMetal = 4, kGlass = 5, kUber = 6 };
// mirror: Kr travels in Ks.  substrate: roughness = uroughness, sigma = vroughness.  metal: Kd = eta, Ks = k, roughness / sigma likewise.  glass: Kd = Kt, Ks = Kr, roughness = eta (index), sigma = uroughness, Kr[0] = vroughness (both 0: smooth).
enum LightType { kPointLight = 0, kDistantLight = 1, kDiffuseAreaLight = 2, kInfiniteLight = 3 };
enum ShapeKind { kTriangleMesh = 0, kSphere = 1 };
enum LightStrategy { kUniform = 0, kPower = 1, kSpatial = 2 };

struct MaterialDesc {
    int32_t type;
    float Kd[3]; float sigma; float Ks[3]; float roughness; int32_t remapRoughness;
    int32_t KdTex, KsTex;     // index into SceneModel::textures when the parameter is an image texture, else -1
    // uber (materials/uber.cpp; roughness = uroughness, sigma = vroughness): what the other materials do not have
    float Kr[3], Kt[3], opacity[3], eta;
    int32_t opacityTex;       // uber: image texture of "opacity" (materials/uber.cpp:53), else -1
};

// ImageTexture<RGBSpectrum, Spectrum> (textures/imagemap.h) with its finished MIPMap (core/mipmap.h): the pyramid
// (level 0 = the power-of-two image, (0,0) = lower left), filter choice and the UVMapping2D parameters.
enum ImageWrapMode { kWrapRepeat = 0, kWrapBlack = 1, kWrapClamp = 2 };
struct MipLevel { int32_t w = 0, h = 0; std::vector<float> rgb; };
struct TextureDesc {
    std::vector<MipLevel> levels;
    int32_t trilinear = 0; float maxAniso = 8.f; int32_t wrap = kWrapRepeat;
    float su = 1.f, sv = 1.f, du = 0.f, dv = 0.f;       // UVMapping2D (core/texture.cpp:93-99)
    float weightLut[128];                                // MIPMap::weightLut
    // What the pyramid was built from (ReadImage's texels, top row first, as 8-bit values when the file held 8-bit channels, and
    // ImageTexture::GetTexture's conversion parameters): lets the baked container carry the image instead of its float
    // pyramid (version 6: 3 bytes per source texel instead of 16 per power-of-two texel) and rebuild the levels at load.
    int32_t srcW = 0, srcH = 0; float srcScale = 1.f; int32_t srcGamma = 0, srcFlipY = 1;
    std::vector<uint8_t> src8; std::vector<float> srcF;   // one of them: 3 * srcW * srcH values; both empty: no source kept
};

struct MeshData {             // world space, as TriangleMesh holds it (shapes/triangle.cpp:54-92)
    std::vector<int32_t> indices;   // 3 per triangle
    std::vector<float> P, N, UV, S; // 3/3/2/3 floats per vertex; N/UV/S may be empty
    uint32_t nTris() const { return (uint32_t)(indices.size() / 3); }
    uint32_t nVerts() const { return (uint32_t)(P.size() / 3); }
};
struct SphereData {
    mat4 objectToWorld, worldToObject;
    float radius, zMin, zMax, thetaMin, thetaMax, phiMax;
};
struct ShapeDesc {
    int32_t kind, material, areaLight, reverseOrientation, transformSwapsHandedness;
    int32_t object = -1;  // >= 0: defined between ObjectBegin/ObjectEnd (core/api.cpp:1752-1774), reachable through instances only
    MeshData mesh;        // kind == kTriangleMesh
    SphereData sphere;    // kind == kSphere
    uint32_t nPrims() const { return kind == kTriangleMesh ? mesh.nTris() : 1u; }
};
struct LightDesc {
    int32_t type;
    float pos[3];     // point: world position; distant: normalised world direction
    float I[3];       // point: I; distant: L; area: Lemit (already multiplied by "scale")
    int32_t shape;    // area: index into shapes
    int32_t twoSided;
    // infinite (lights/infinite.cpp): radiance map = textures[texture] (texels already times L * scale, NOT flipped in y: the
    // light indexes its map as read; a constant light has a 1x1 map), and the light <-> world transform
    int32_t texture;
    mat4 lightToWorld, worldToLight;
};
// ObjectInstance (core/api.cpp:1778-1820): a TransformedPrimitive over the object's primitives
struct InstanceDesc { int32_t object; mat4 instanceToWorld, worldToInstance; };
// renderOptions->primitives in creation order: a shape's primitives (kind 0) or one instance (kind 1)
struct TopItem { int32_t kind; uint32_t index; };
struct RenderOptions {
    int32_t xres = 1280, yres = 720;
    float crop[4] = {0, 1, 0, 1};                 // x0 x1 y0 y1
    float filterRadius[2] = {0.5f, 0.5f}; int32_t filterType = 0;   // box
    float filmScale = 1.f, maxSampleLuminance = HPRT_INF;
    float fov = 90.f, lensRadius = 0.f, focalDistance = 1e6f;
    float screenWindow[4] = {-1, 1, -1, 1};       // x0 x1 y0 y1
    float shutterOpen = 0.f, shutterClose = 1.f;
    mat4 cameraToWorld, worldToCamera;
    int32_t spp = 16, samplePixelCenter = 0;
    int32_t maxDepth = 5; float rrThreshold = 1.f; int32_t lightStrategy = kSpatial;
    int32_t maxNodePrims = 4, isectCost = 8, travCost = 1;
    std::string filename = "pbrt.exr", accelerator = "bvh", integrator = "path", sampler = "halton";
};
struct SceneModel {
    RenderOptions opt;
    std::vector<MaterialDesc> materials;
    std::vector<ShapeDesc> shapes;
    std::vector<LightDesc> lights;
    std::vector<TextureDesc> textures;
    uint32_t nObjects = 0;                // object ids are 0..nObjects-1 (ShapeDesc::object)
    std::vector<InstanceDesc> instances;
    std::vector<TopItem> top;             // what the top-level aggregate holds, in creation order
    std::vector<std::string> warnings;
    // primitives of the top-level aggregate
    uint64_t totalPrims() const { uint64_t n = 0; for (auto &t : top) n += t.kind == 0 ? shapes[t.index].nPrims() : 1u; return n; }
};

// texture_io.cpp: ReadImage (core/imageio.cpp:60-79; .tga .png .pfm), rows top to bottom, RGB floats; and
// ImageTexture::GetTexture + MIPMap::MIPMap (textures/imagemap.cpp:52-97, core/mipmap.h:113-201)
bool ReadImageFile(const std::string &path, int *w, int *h, std::vector<float> *rgb, std::string *err);
void BuildMipMap(int w, int h, const std::vector<float> &rgb, float scale, bool gamma, TextureDesc *tex, bool flipY = true);
// keeps ReadImage's texels with the texture (as bytes when every value is k / 255.f) so that SaveBakedScene can store them compactly
void KeepTextureSource(int w, int h, const std::vector<float> &rgb, float scale, bool gamma, bool flipY, TextureDesc *tex);
// rebuilds the levels of a texture loaded in source form
void RebuildFromSource(TextureDesc *tex);

bool SaveBakedScene(const SceneModel &sc, const std::string &path, std::string *err, bool compactTextures = false);
bool LoadBakedScene(const std::string &path, SceneModel *sc, std::string *err);
// pbrt front-end (pbrt_frontend.cpp): parses the directive subset of SURVEY.md §8(f)-2
bool ParsePbrtFile(const std::string &path, const std::map<std::string, std::string> &subst, SceneModel *sc,
                   std::string *err);
bool ParsePbrtString(const std::string &text, const std::string &baseDir,
                     const std::map<std::string, std::string> &subst, SceneModel *sc, std::string *err);

}  // namespace hprt
