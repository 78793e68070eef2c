// hprt device side — kernel parameter blocks and launcher prototypes (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "dev_shading.h"

namespace hprt {

struct RayPlanes { float *ox, *oy, *oz, *dx, *dy, *dz, *tmax; };      // SoA, indexed by slot
struct HitPlanes { float *t; int32_t *prim; float *b0, *b1, *b2; };

// Per-path state in HBM, one float/uint plane per field (coalesced when consecutive
// lanes hold consecutive slots).
struct PathPlanes {
    RayPlanes ray;          // current path segment
    HitPlanes hit;          // its closest hit
    float *betaR, *betaG, *betaB, *LR, *LG, *LB;
    uint32_t *state;        // sampler dimension (bits 0-7) | bounces (bits 8-15)
    RayPlanes sh;           // shadow ray of the light sample
    RayPlanes mis;          // BSDF-sampled MIS ray (tmax plane unused: Infinity)
    HitPlanes misHit;
    uint8_t *occluded;
    float *pendLightR, *pendLightG, *pendLightB, *pendMisR, *pendMisG, *pendMisB, *pendBetaR, *pendBetaG, *pendBetaB, *pendPdf;
    uint32_t *pendInfo;     // light number | bit30 shadow ray issued | bit31 MIS ray issued
};
struct QueueSet {
    uint32_t *next, *shadow, *mis, *resolve;
    uint32_t *nextCount, *shadowCount, *misCount, *resolveCount;
};
struct BinSet { uint32_t *q[3]; uint32_t *count; };   // material bins: 0 matte, 1 plastic, 2 generic (count[3])
struct RenderParams {
    DevCamera cam;
    DevHalton hal;
    const uint32_t *pixelXY;        // x | y << 16 of every local pixel, tile by tile
    const uint64_t *pixelOffset;    // Halton index of sample 0: offsetForCurrentPixel (samplers/halton.cpp:101-118)
    uint32_t nPix;
    int32_t maxDepth;
    float rrThreshold;
};
struct FilmGeom {
    int32_t cx0, cy0, cx1, cy1;     // croppedPixelBounds
    int32_t sx0, sy0, sx1, sy1;     // sample bounds
    float rx, ry;                   // filter radius
    float maxSampleLuminance;
};
struct IrregularSample { uint32_t pix, sample; int16_t x0, x1, y0, y1; };
// Samples that also land in pixels other than their own, grouped per destination.
struct FilmExtras {
    // same-tile: CSR over local pixels; entries sorted in the tile loop's order, pre first
    const uint32_t *ownBegin; const uint32_t *ownSrcPix; const uint32_t *ownSample; const uint8_t *ownIsPre;
    // other-tile: destinations -> groups (one per source tile) -> entries
    uint32_t nForeignDest;
    const uint32_t *foreignDestFilmIndex; const uint32_t *foreignDestBegin; const uint32_t *foreignGroupBegin;
    const uint32_t *foreignSrcPix; const uint32_t *foreignSample;
};

void LaunchTrace(hipStream_t st, const DevScene &sc, bool anyHit, bool count, const uint32_t *queue, const uint32_t *countPtr,
                 uint32_t countImm, uint32_t gridItems, const RayPlanes &rays, const HitPlanes &hits, uint8_t *occ,
                 DevCounters *counters, uint32_t *workCounter);
void LaunchGenerate(hipStream_t st, const DevScene &sc, const RenderParams &rp, const PathPlanes &ps, uint32_t s0, uint32_t nSlots);
void LaunchBin(hipStream_t st, const DevScene &sc, const PathPlanes &ps, const uint32_t *queue, const uint32_t *countPtr,
               uint32_t countImm, uint32_t gridItems, int32_t maxDepth, const BinSet &bins);
void LaunchShade(hipStream_t st, int mode, const DevScene &sc, const RenderParams &rp, const PathPlanes &ps, const uint32_t *queue,
                 const uint32_t *countPtr, uint32_t countImm, uint32_t gridItems, uint32_t s0, const QueueSet &q, const BinSet &bins);
void LaunchResolve(hipStream_t st, const DevScene &sc, const PathPlanes &ps, const uint32_t *queue, const uint32_t *countPtr,
                   uint32_t gridItems);
void LaunchStoreRadiance(hipStream_t st, const PathPlanes &ps, float *LallR, float *LallG, float *LallB, uint32_t nPix, uint32_t s0,
                         uint32_t nSlots);
void LaunchFindIrregular(hipStream_t st, const DevScene &sc, const RenderParams &rp, const FilmGeom &fg, uint32_t spp, uint32_t *count,
                         uint32_t capacity, IrregularSample *out);
void LaunchFilmOwn(hipStream_t st, const RenderParams &rp, const FilmGeom &fg, const float *LallR, const float *LallG, const float *LallB,
                   uint32_t spp, const FilmExtras &ex, float *film);
void LaunchFilmForeign(hipStream_t st, const RenderParams &rp, const FilmGeom &fg, const float *LallR, const float *LallG,
                       const float *LallB, const FilmExtras &ex, float *film);

}  // namespace hprt
