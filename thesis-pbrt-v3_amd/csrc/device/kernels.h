// hprt device side — kernel parameter blocks and launcher prototypes (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "dev_shading.h"

namespace hprt {

// Path data lives in HBM as STREAMS of 16-byte words: a lane reads or writes a whole word with
// one request, and a shading pass writes its results at its own thread index, so consecutive
// lanes touch consecutive words.  Index spaces are re-made every bounce: the paths that a bounce
// shades are numbered 0..n-1 in shading order (matte bin, plastic bin, generic bin) and
// everything that bounce produces — the next path segment, the shadow and MIS rays, the pending
// light terms — is stored under that number; queues list the numbers that need a given pass
// (in ascending runs, one per workgroup), so gathers stay almost sequential however thin the
// paths become.
struct RayStream { float4 *a, *b; };                 // a = {o.xyz, tMax}   b = {d.xyz, aux}
struct HitStream { float4 *a; float2 *b; };           // a = {t, prim, b0, b1}   b = {b2, instance or -1}; b may be null
struct PathStream {
    RayStream ray;          // current path segment; ray.b.w = sampler dimension (bits 0-15) | bounces (bits 16-30) | the segment left a specular lobe (bit 31)
    float4 *beta;           // {beta.rgb, path id}   path id = sampleInBatch * nPix + pixel
    float4 *L;              // {L.rgb, w}: w = 0 if the path ends at this vertex, else its etaScale (1 until it crosses a dielectric boundary)
};
// What a shading pass leaves for the rest of its bounce, indexed like its output PathStream
struct VertexStreams {
    RayStream shadow; uint8_t *occluded;             // shadow ray of the light sample, its any-hit result
    RayStream mis; HitStream misHit;                 // BSDF-sampled MIS ray, its closest hit (b2 unused)
    float4 *pendLight;      // {light-sampling term rgb, info}   info: light number | bit30 shadow ray issued | bit31 MIS ray issued
    float4 *pendMis;        // {BSDF-sampling term rgb, light pick pdf}
    float4 *pendBeta;       // {beta before this vertex rgb, path id}
};
struct QueueSet {
    uint32_t *next, *shadow, *mis, *resolve;
    uint32_t *nextCount, *shadowCount, *misCount, *resolveCount;
};
// material bins (BIN_*, dev_scene.h): 0 matte, 1 plastic, 2 generic, 3 generic on an image-textured material (the variant compiled
// with the MIPMap lookups: scenes with textures only), 4 substrate.  count[k * BIN_STRIDE], k = 0..4: sizes; k = 5: size of bin 2
// before the specialised variants deferred vertices to it (counters 256 bytes apart: atomics on different bins do not share a line);
// k = 6, 7: the retry lists; aux[k] = output index of a deferred entry k of bin 2.  Output indices: bin 0 occupies [0, n0), then
// bins 1, 4, 2 (as binned) and 3.
enum : uint32_t { BIN_STRIDE = 64u };
// retry[b - 2], b = 2, 3: {stream index, output index} pairs of vertices whose light-distribution voxel was not there yet
// (on-demand SpatialLightDistribution): shaded again by bin b's variant after the voxels have been filled; count[(6 + b - 2) * BIN_STRIDE]
struct BinSet { uint32_t *q[5]; uint32_t *aux; uint32_t *count; uint2 *retry[2]; };
struct RenderParams {
    DevCamera cam;
    DevHalton hal;
    const uint32_t *pixelXY;        // x | y << 16 of every local pixel, tile by tile
    const uint64_t *pixelOffset;    // Halton index of sample 0: offsetForCurrentPixel (samplers/halton.cpp:101-118)
    uint32_t nPix;
    int32_t maxDepth;
    float rrThreshold;
    float invSqrtSpp;               // 1 / sqrt(samplesPerPixel): ray differential scale (image textures only)
    int32_t cullMis;                // do not trace rays whose result provably changes nothing (k_shade): a BSDF-sampled light ray that
                                    // cannot reach its emitter, the segment behind the last vertex of a path
};
struct FilmGeom {
    int32_t cx0, cy0, cx1, cy1;     // croppedPixelBounds
    int32_t sx0, sy0, sx1, sy1;     // sample bounds
    float rx, ry;                   // filter radius
    float maxSampleLuminance;
};
struct IrregularSample { uint32_t pix, sample; int16_t x0, x1, y0, y1; };
// k_generate forms every camera sample's film position anyway: given this it also lists the irregular ones of its batch (count is
// cumulative over the batches of a render; entries beyond capacity are counted, not stored — the host then falls back to k_find_irregular)
struct IrregularSink { FilmGeom fg; uint32_t *count; uint32_t capacity; IrregularSample *out; };
// Samples that also land in pixels other than their own, grouped per destination.
struct FilmExtras {
    // same-tile: CSR over local pixels; entries sorted in the tile loop's order, pre first
    const uint32_t *ownBegin; const uint32_t *ownSrcPix; const uint32_t *ownSample; const uint8_t *ownIsPre;
    // other-tile: destinations -> groups (one per source tile) -> entries
    uint32_t nForeignDest;
    const uint32_t *foreignDestFilmIndex; const uint32_t *foreignDestBegin; const uint32_t *foreignGroupBegin;
    const uint32_t *foreignSrcPix; const uint32_t *foreignSample;
};

// One cross-tile film contribution (HprtFilmRecord of include/hprt.h): the fold of one source tile's samples that land in
// film pixel `dest` of another tile
struct FilmRecord { uint32_t dest, srcTile; float xyz[3]; float w; };

bool WideWalkInUse(const DevScene &sc);      // plain renders of this scene take k_walk4 (the leaf-exact wide walk) rather than k_trace
void LaunchTrace(hipStream_t st, const DevScene &sc, bool anyHit, bool count, const uint32_t *queue, const uint32_t *countPtr,
                 uint32_t countImm, uint32_t gridItems, const RayStream &rays, const HitStream &hits, uint8_t *occ,
                 DevCounters *counters, uint32_t *workCounter, uint4 *rayStats = nullptr);
void LaunchPixelStats(hipStream_t st, const uint4 *rayStats, const float4 *ids, const uint32_t *queue, const uint32_t *countPtr,
                      uint32_t countImm, uint32_t gridItems, uint32_t nPix, bool anyHit, uint32_t *pix);
void LaunchPixelStatsToFilm(hipStream_t st, const uint32_t *pix, const uint32_t *pixelXY, uint32_t nPix, uint32_t spp, int cx0, int cy0, int width,
                            unsigned long long *out7);
void LaunchGenerate(hipStream_t st, const DevScene &sc, const RenderParams &rp, const PathStream &out, uint32_t s0, uint32_t nSlots,
                    const IrregularSink *irr = nullptr);
void LaunchBin(hipStream_t st, const DevScene &sc, const PathStream &in, const HitStream &hit, const uint32_t *queue,
               const uint32_t *countPtr, uint32_t countImm, uint32_t gridItems, int32_t maxDepth, int32_t bounces, const BinSet &bins,
               float4 *Lfinal);
void LaunchShade(hipStream_t st, int mode, const DevScene &sc, const RenderParams &rp, const PathStream &in, const HitStream &hit,
                 uint32_t gridItems, uint32_t s0, const PathStream &out, const VertexStreams &vs, const QueueSet &q,
                 const BinSet &bins, float4 *Lfinal, bool firstBounce, bool retryPass = false);
void LaunchResolve(hipStream_t st, const DevScene &sc, const VertexStreams &vs, float4 *L, float4 *Lfinal, const uint32_t *queue,
                   const uint32_t *countPtr, uint32_t gridItems);
void LaunchStoreRadiance(hipStream_t st, const float4 *Lfinal, float *LallR, float *LallG, float *LallB, uint32_t nPix, uint32_t s0,
                         uint32_t nSlots);
// the *_device entry points of include/hprt.h keep their plane layouts: [7][n] rays in, t / prim / [3][n] barycentrics out
void LaunchPackRays(hipStream_t st, const float *rays7, uint32_t n, const RayStream &out);
void LaunchUnpackHits(hipStream_t st, const HitStream &hits, uint32_t n, float *t, int32_t *prim, float *bary3);   // instance ids stay in hits.b
void LaunchFindIrregular(hipStream_t st, const DevScene &sc, const RenderParams &rp, const FilmGeom &fg, uint32_t spp, uint32_t *count,
                         uint32_t capacity, IrregularSample *out);
void LaunchFilmOwn(hipStream_t st, const RenderParams &rp, const FilmGeom &fg, const float *LallR, const float *LallG, const float *LallB,
                   uint32_t spp, const FilmExtras &ex, float *film);
void LaunchFilmForeign(hipStream_t st, const RenderParams &rp, const FilmGeom &fg, const float *LallR, const float *LallG,
                       const float *LallB, const FilmExtras &ex, float *film);
void LaunchFilmForeignExport(hipStream_t st, const RenderParams &rp, const FilmGeom &fg, const float *LallR, const float *LallG,
                             const float *LallB, const FilmExtras &ex, uint32_t nGroups, const uint32_t *groupDest, const uint32_t *groupTile,
                             FilmRecord *out);
// SpatialLightDistribution for every voxel: ri = RadicalInverse(0..4, i), i < 128, as [5][128] floats on the device
void LaunchVoxelDistributions(hipStream_t st, const DevScene &sc, const float *ri, uint32_t nVox, float *func, float *cdf, float *funcInt);
// on-demand mode: the `n` voxels listed in sc.voxRequest get rows rowBase .. rowBase + n - 1 (voxSlot is updated), their distributions are computed
void LaunchVoxelFill(hipStream_t st, const DevScene &sc, const float *ri, uint32_t n, uint32_t rowBase, float *func, float *cdf, float *funcInt);
void LaunchStreamCopy(hipStream_t st, const float4 *src, float4 *dst, size_t n);
void LaunchGatherProbe(hipStream_t st, const uint4 *records, uint32_t log2Records, int itersPerLane, uint32_t blocks, uint32_t *sink);
// diagnostics: the rays a queue lists, as [7][cap] planes (the layout of the *_device entry points)
void LaunchCaptureRays(hipStream_t st, const uint32_t *queue, uint32_t n, const RayStream &rays, float *out7, uint32_t cap);
void LaunchFilmApplyRecords(hipStream_t st, const FilmRecord *rec, const uint32_t *destBegin, uint32_t nDest, float *film);

}  // namespace hprt
