// hprt — host-side state of a device scene, shared by capi_device.hip (upload, trace, render) and
// capi_gather.hip (multi-GPU film gather).  Library-internal.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/hprt.h"
#include "device/kernels.h"
#include "hprt_internal.h"

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return hprt::SetError(HPRT_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__));   \
    } while (0)

namespace hprt {

struct DevBuf {
    void *p = nullptr; size_t bytes = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) {
        if (p && bytes >= n) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        if (n == 0) return hipSuccess;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    template <typename T> T *as() const { return (T *)p; }
};
template <typename T> hipError_t upload(DevBuf &b, const std::vector<T> &v) {
    hipError_t e = b.alloc(std::max<size_t>(v.size() * sizeof(T), 16));
    if (e != hipSuccess || v.empty()) return e;
    return hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
}

}  // namespace hprt

struct HprtScene;
namespace hprt {
// Scope of one call on a scene: takes the scene's mutex and orders the call's stream behind the last asynchronous call.
struct SceneCall {
    std::unique_lock<std::mutex> lk; HprtScene *s; hipStream_t st;
    SceneCall(HprtScene *scene, hipStream_t stream);
    void leave_async();      // the call returns with work still queued on `st`: later calls must wait for it
};
}  // namespace hprt

struct HprtScene {
    int device = 0;
    // One call in flight per scene (include/hprt.h "Concurrency"): the work counter, the stream copies of the *_device calls, the
    // deep-stack area and the render workspace belong to the scene, not to a call.  Host threads are serialised by `mu`; the
    // asynchronous *_device calls leave `lastUse` behind on their stream and every later call makes ITS stream wait for it, so
    // calls on different streams run one after the other on the device too.
    std::mutex mu;
    hipEvent_t lastUse = nullptr; bool lastUsePending = false;
    hprt::DevScene dev;
    hprt::DevBuf textures, mipLevels, texels, weightLut;
    hprt::DevBuf nodes, tris, primVtx, primN, vUV, vS, shapes, materials, lights, spheres, instances, topEntry, topEntryWide, lightFunc, lightCdf, perms, primes, primeSums, primeMagic;
    hprt::DevBuf envLights, envData;      // infinite lights: DevEnvLight table and their Distribution2D tables
    hprt::DevBuf wide, leafBox;           // the leaf-exact walk structure (wide_bvh.h)
    hprt::DevBuf counters, workCounter, deepStack;
    // hprt_debug_capture_rays (tools/sort_experiment.py): the next render copies the rays one bounce queues into a caller buffer
    struct Capture { int bounce = -1, kind = 0; float *out7 = nullptr; size_t cap = 0, n = 0; } capture;
    int poisonByte = -1;      // hprt_debug_poison_workspace (tests): fill every stream, queue and stack with this byte before each render
    hprt::DevBuf voxFunc, voxCdf, voxFuncInt, voxRi;      // SpatialLightDistribution tables (lightsamplestrategy "spatial")
    hprt::DevBuf voxSlot, voxRequest, voxRequestCount, retryQueues;      // on-demand mode: voxel -> table row, the request list, the vertices to shade again
    uint32_t voxRows = 0, voxRowsUsed = 0, nVoxels = 0;                   // rows the tables can hold / hold
    hprt::DevBuf rayStats, pixelStatsLocal, pixelStatsFilm; bool pixelStatsValid = false;   // HPRT_RENDER_PIXEL_STATS
    // render-time state
    hprt::DevBuf planes;                                    // backing store of the path streams (Workspace)
    hprt::DevBuf apiRays, apiHits;                          // stream copies of the plane-layout arguments of the *_device calls
    hprt::DevBuf queues, queueCounts;
    hprt::DevBuf pixelXY, pixelOffset, Lall, film, irregular, irregularCount;
    hprt::DevBuf exOwnBegin, exOwnSrc, exOwnSample, exOwnPre, exFDest, exFDestBegin, exFGroupBegin, exFSrc, exFSample;
    // HPRT_RENDER_EXPORT_FOREIGN: the cross-tile film contributions of the last render, one record per (destination
    // pixel, source tile), sorted by both; applied by hprt_film_gather on the root in the single-GPU order
    hprt::DevBuf foreignRecords, exGroupDest, exGroupTile; uint32_t nForeignRecords = 0; bool foreignExported = false;
    int filmW = 0, filmH = 0;
    float *lastFilm = nullptr;                        // the device buffer the last hprt_render wrote: the caller's, or `film`
    uint32_t *hostCounts = nullptr;                   // pinned
    size_t filmPixels = 0;
    uint32_t nPrims = 0;
    bool hasSubstrateBin = false;                     // some triangle carries BIN_SUBSTRATE: the substrate shading variant is launched
    ~HprtScene() { if (hostCounts) (void)hipHostFree(hostCounts); if (lastUse) (void)hipEventDestroy(lastUse); }
};

inline hprt::SceneCall::SceneCall(HprtScene *scene, hipStream_t stream) : lk(scene->mu), s(scene), st(stream) {
    if (s->lastUsePending && s->lastUse) (void)hipStreamWaitEvent(st, s->lastUse, 0);
}
inline void hprt::SceneCall::leave_async() {
    if (!s->lastUse && hipEventCreateWithFlags(&s->lastUse, hipEventDisableTiming) != hipSuccess) { s->lastUse = nullptr; (void)hipStreamSynchronize(st); s->lastUsePending = false; return; }
    s->lastUsePending = hipEventRecord(s->lastUse, st) == hipSuccess;
    if (!s->lastUsePending) (void)hipStreamSynchronize(st);
}
