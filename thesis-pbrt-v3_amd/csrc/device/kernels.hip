// hprt device side — the wavefront path-tracing kernels (gfx950, wave64).
//
// Path data lives in streams of 16-byte words (kernels.h); kernels communicate through
// queues of stream indices built with workgroup-level ballot/prefix compaction (one atomic
// per queue and workgroup).  Per bounce:
//     k_trace<closest>  path rays            BVHAccel::Intersect + Triangle::Intersect
//     k_shade           Li loop body         PathIntegrator::Li (integrators/path.cpp:64-204),
//                                            UniformSampleOneLight/EstimateDirect set-up
//                                            (core/integrator.cpp:86-217)
//     k_trace<any>      shadow rays          BVHAccel::IntersectP (VisibilityTester::Unoccluded)
//     k_trace<closest>  MIS rays             scene.Intersect at core/integrator.cpp:195
//     k_resolve         L += beta * Ld / lightPdf, in bounce order
// and once per batch k_generate (Render loop head, core/integrator.cpp:281-293) and
// k_store_radiance (radiance guards, :300-321).  The film (FilmTile::AddSample /
// MergeFilmTile, core/film.h:130-170, core/film.cpp:118-132) is folded per pixel in
// exact sample order by k_film_*.
//
// Built with -ffp-contract=off: every float operation below is a single IEEE
// rounding in the order the reference's scalar code performs it.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include "dev_shading.h"
#include "kernels.h"
#ifndef HPRT_QUAD_ANY_WAVES
#define HPRT_QUAD_ANY_WAVES 5
#endif
#define HPRT_QUAD_CLOSEST_WAVES 4
#ifndef HPRT_LDS_STACK_QUAD_ANY
#define HPRT_LDS_STACK_QUAD_ANY 12
#endif
#define HPRT_LDS_STACK_INST_ANY 7
#define HPRT_LDS_STACK_INST_CLOSEST 9
#ifndef HPRT_ANY_WAVES
#define HPRT_ANY_WAVES 7
#endif
#ifndef HPRT_CLOSEST_WAVES
#define HPRT_CLOSEST_WAVES 6
#endif
#ifndef HPRT_LDS_STACK_CLOSEST
#define HPRT_LDS_STACK_CLOSEST 12
#endif

namespace hprt {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------
// wave-level queue append: returns the position for lanes with pred, one atomic per wave
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_append(uint32_t *counter, bool pred) {
    const unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return 0u;
    const uint32_t lane = __lane_id();
    const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0u;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    return base + prefix;
}

// Block-level queue append for up to four queues at once.  Global atomics on ONE address
// retire at only ~90 per microsecond on this chip, so appends are aggregated over the
// whole workgroup: ballots per wave, a 16-entry scan in LDS, then one atomic per queue and
// workgroup (1024 lanes) instead of one per wave.  Must be called by every thread of the
// block in uniform control flow.  pos[k] is valid where pred[k] is set.
struct BlockAppendLds { uint32_t waveCount[4][16]; uint32_t base[4]; };
template <int NQ>
__device__ __forceinline__ void block_append(BlockAppendLds *lds, uint32_t *const counter[NQ], const bool pred[NQ], uint32_t pos[NQ]) {
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6, nWaves = (blockDim.x + 63) >> 6;
    unsigned long long mask[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        mask[k] = __ballot(pred[k]);
        if (lane == 0) lds->waveCount[k][wave] = (uint32_t)__popcll(mask[k]);
    }
    __syncthreads();
    if (threadIdx.x < NQ) {
        const int k = threadIdx.x;
        uint32_t total = 0;
        for (uint32_t w = 0; w < nWaves; ++w) { const uint32_t c = lds->waveCount[k][w]; lds->waveCount[k][w] = total; total += c; }
        lds->base[k] = total ? atomicAdd(counter[k], total) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NQ; ++k)
        pos[k] = lds->base[k] + lds->waveCount[k][wave] +
                 __builtin_amdgcn_mbcnt_hi((uint32_t)(mask[k] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask[k], 0u));
}

__device__ __forceinline__ void wave_count_add(DevCounters *c, bool anyHit, const TraceCount &t) {
    // per-wave reduction, then one atomic per counter per wave
    unsigned int f = t.fetched, e = t.entered, tr = t.tri, sp = t.sphere;
    for (int off = 32; off > 0; off >>= 1) {
        f += __shfl_down(f, off); e += __shfl_down(e, off); tr += __shfl_down(tr, off); sp += __shfl_down(sp, off);
    }
    if (__lane_id() == 0) {
        if (!anyHit) { atomicAdd(&c->nodesFetched, (unsigned long long)f); atomicAdd(&c->nodesEntered, (unsigned long long)e);
                       atomicAdd(&c->triTests, (unsigned long long)tr); atomicAdd(&c->sphereTests, (unsigned long long)sp); }
        else { atomicAdd(&c->nodesFetchedP, (unsigned long long)f); atomicAdd(&c->nodesEnteredP, (unsigned long long)e);
               atomicAdd(&c->triTestsP, (unsigned long long)tr); atomicAdd(&c->sphereTestsP, (unsigned long long)sp); }
    }
}

// ---------------------------------------------------------------------------
// k_trace: persistent wavefronts with dynamic ray fetch over the child-pair layout.
//
// Incoherent rays have very uneven traversal lengths (killeroo-simple bounce rays: median
// 7 nodes, mean 25, per-64-ray maximum 78), so a one-ray-per-lane kernel keeps ~1/3 of a
// wavefront's lanes busy.  Here a wave owns 64 lane slots for its whole life: whenever
// fewer than `refillBelow` lanes still have a ray, the idle lanes draw the next rays
// from the queue (ballot + mbcnt prefix over a per-wave chunk) and join the walk.
//
// A step reads one DevPair — an interior node with the bounds of BOTH children — and runs
// the reference's slab test (core/geometry.h:1754-1780) on each.  The reference
// (accelerators/bvh.cpp:363-394) pushes the far child unconditionally and tests its bounds
// when it is popped, against the tMax of that later moment.  The slab part of that test
// does not depend on tMax, so here it is evaluated at the parent: a far child whose slabs
// miss is never pushed, one whose slabs hit is pushed together with its entry distance
// tMin, and the pop re-applies exactly the one tMax-dependent comparison of the reference
// (tMin < ray.tMax) with the tMax current at the pop.  The set and order of nodes entered
// and of primitives tested — hence every hit, t and barycentric — is the reference's;
// popped-and-culled nodes simply cost no memory access.  With COUNT the far child is always
// pushed (tMin = +inf when its slabs miss) so that "fetched" counts the reference's pops.
//
// Lane state is one word, `cur`: >= 0 an interior pair to step, < 0 a parked primitive
// (~index; the last primitive of a leaf carries TAG_LAST), REF_NONE when the ray is done.
// The wave alternates two wave-uniform phases ("while-while"): pair steps until enough
// lanes are parked, then the parked primitives together.  A lane moves on only after its
// parked leaf has been processed, so every later test sees the shrunken tMax exactly as in
// the reference.
//
// queue == nullptr means "slot = ray index".  Closest hit writes {t, prim (ordered index or
// -1), b0, b1} and b2; any hit writes one byte.
// ---------------------------------------------------------------------------
// Scheduling knobs of the persistent walk (defaults from a per-ray trace simulation of
// killeroo-simple bounce rays; overridable through HPRT_TRACE_TUNE / HPRT_TRACE_TUNE_ANY =
// "refillBelow,parkLimit,stepLimit,sphereLimit,primMin").
struct TraceTune { int refillBelow, parkLimit, stepLimit, sphereLimit, primMin; };
static TraceTune DefaultTraceTune(bool anyHit) {
    // closest-hit and any-hit rays want different schedules: shadow rays mostly cross the scene unoccluded, with
    // few primitive tests each, so their tests should not wait for company (full-frame sweeps, tools/sweep_bench.sh)
    // (any hit: refill below 40 busy lanes since the seven-wave kernels, tools/sweep_tune3.sh: atrium shadow rays +5 %, the others +-0)
    TraceTune t = anyHit ? TraceTune{40, 24, 10, 4, 3} : TraceTune{52, 24, 6, 4, 8};      // (sphereLimit: flat between 1 and 16 since the pre-test)
    if (const char *e = getenv(anyHit ? "HPRT_TRACE_TUNE_ANY" : "HPRT_TRACE_TUNE"))
        sscanf(e, "%d,%d,%d,%d,%d", &t.refillBelow, &t.parkLimit, &t.stepLimit, &t.sphereLimit, &t.primMin);
    return t;
}

// Bounds3::IntersectP(ray, invDir, dirIsNeg) (core/geometry.h:1754-1780) split into its tMax-independent part (returned)
// and the entry distance for the `tMin < ray.tMax` part.  Branch-free: the reference's early returns only skip work,
// they do not change the values that the later comparisons see.  (One child at a time: the form the kernels with the
// quadric code use — the packed form below needs registers that their 128 do not have.)
__device__ __forceinline__ bool slab_test(float lox, float hix, float loy, float hiy, float loz, float hiz, vec3 ro, vec3 invDir,
                                          bool negX, bool negY, bool negZ, float robust, float *tEntry) {
    float tMin = ((negX ? hix : lox) - ro.x) * invDir.x;
    float tMax = ((negX ? lox : hix) - ro.x) * invDir.x;
    const float tyMin = ((negY ? hiy : loy) - ro.y) * invDir.y;
    float tyMax = ((negY ? loy : hiy) - ro.y) * invDir.y;
    tMax *= robust; tyMax *= robust;
    bool ok = !(tMin > tyMax || tyMin > tMax);
    if (tyMin > tMin) tMin = tyMin;
    if (tyMax < tMax) tMax = tyMax;
    const float tzMin = ((negZ ? hiz : loz) - ro.z) * invDir.z;
    float tzMax = ((negZ ? loz : hiz) - ro.z) * invDir.z;
    tzMax *= robust;
    ok = ok && !(tMin > tzMax || tzMin > tMax);
    if (tzMin > tMin) tMin = tzMin;
    if (tzMax < tMax) tMax = tzMax;
    *tEntry = tMin;
    return ok && (tMax > 0);
}

// Bounds3::IntersectP(ray, invDir, dirIsNeg) (core/geometry.h:1754-1780) for BOTH children of a pair at once, split into
// its tMax-independent part (returned per child) and the entry distances for the `tMin < ray.tMax` part.  n* hold the
// bounds the reference selects with dirIsNeg for tMin (pMin, or pMax where the ray runs backwards), f* those for tMax,
// as {child 0, child 1}: the arithmetic runs as packed two-float operations (v_pk_add_f32 / v_pk_mul_f32: one IEEE
// rounding per component and operation, no fusing), the comparisons and selects per child.  Branch-free: the
// reference's early returns only skip work, they do not change the values that the later comparisons see.
__device__ __forceinline__ void slab_test_pair(f32x2 nx, f32x2 fx, f32x2 ny, f32x2 fy, f32x2 nz, f32x2 fz, float rox, float roy, float roz,
                                               float ivx, float ivy, float ivz, float robust, bool *s0, bool *s1, float *t0, float *t1) {
    // (scalars, not vec3: hipcc otherwise builds the {y, z} operand pair through a scratch store and reload of the struct)
    f32x2 tMin = (nx - rox) * ivx;
    f32x2 tMax = (fx - rox) * ivx;
    const f32x2 tyMin = (ny - roy) * ivy;
    f32x2 tyMax = (fy - roy) * ivy;
    tMax = tMax * robust; tyMax = tyMax * robust;
    const f32x2 tzMin = (nz - roz) * ivz;
    f32x2 tzMax = (fz - roz) * ivz;
    tzMax = tzMax * robust;
    bool ok0 = !(tMin.x > tyMax.x || tyMin.x > tMax.x), ok1 = !(tMin.y > tyMax.y || tyMin.y > tMax.y);
    if (tyMin.x > tMin.x) tMin.x = tyMin.x;
    if (tyMin.y > tMin.y) tMin.y = tyMin.y;
    if (tyMax.x < tMax.x) tMax.x = tyMax.x;
    if (tyMax.y < tMax.y) tMax.y = tyMax.y;
    ok0 = ok0 && !(tMin.x > tzMax.x || tzMin.x > tMax.x); ok1 = ok1 && !(tMin.y > tzMax.y || tzMin.y > tMax.y);
    if (tzMin.x > tMin.x) tMin.x = tzMin.x;
    if (tzMin.y > tMin.y) tMin.y = tzMin.y;
    if (tzMax.x < tMax.x) tMax.x = tzMax.x;
    if (tzMax.y < tMax.y) tMax.y = tzMax.y;
    *t0 = tMin.x; *t1 = tMin.y;
    *s0 = ok0 && (tMax.x > 0); *s1 = ok1 && (tMax.y > 0);
}

// Diagnostics (HPRT_TRACE_PROFILE=1, tools/trace_profile.py): per-phase cycles and lane occupancy,
// summed over all waves.  [0] total [1] refill [2] pair phase [3] primitive phase [4] quadric
// batches [5] pair iterations [6] pair lanes [7] primitive iterations [8] primitive lanes
// [9] refills [10] refilled lanes [11] quadric batches [12] quadric lanes [13] waves [14] stack pushes [15] pushes beyond the LDS entries.
__device__ unsigned long long g_traceProf[32];   // [0..15] closest hit, [16..31] any hit
// HPRT_SHADE_PROF (variant builds only, tools/build_variant.sh): wave clocks between program points of k_shade, per MODE
#ifdef HPRT_SHADE_PROF
__device__ unsigned long long g_shadeProf[4 * 8];
__device__ unsigned long long g_shadeLanes[4 * 8 * 2];      // per MODE and mark: sum of active lanes, number of times reached
#define SP_MARK(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = clock64(); spT[k] += t_ - spLast; spLast = t_; \
                        const unsigned long long m_ = __ballot(1); if (__lane_id() == (uint32_t)(__ffsll((long long)m_) - 1)) { atomicAdd(&g_shadeLanes[(MODE * 8 + k) * 2], (unsigned long long)__popcll(m_)); atomicAdd(&g_shadeLanes[(MODE * 8 + k) * 2 + 1], 1ull); } } while (0)
#else
#define SP_MARK(k) do { } while (0)
#endif

// MODE 0: plain; 1: work counters (HprtRenderStats); 2: phase profile (diagnostics only).
// INST: the scene has object instances (two-level walk); without them that code and its registers are compiled out.
// QUAD: the scene has quadrics (spheres).  Their interval-arithmetic test is a call of 99 VGPRs that every value the walk
// keeps must sit above; triangle-only scenes (the Sponza-class and living-room workloads) get the kernel without it.
template <bool ANY_HIT, int MODE, bool INST, bool QUAD>
__global__ __launch_bounds__(HPRT_TRACE_BLOCK, MODE == 0 ? (QUAD ? (ANY_HIT ? HPRT_QUAD_ANY_WAVES : HPRT_QUAD_CLOSEST_WAVES) : ANY_HIT ? HPRT_ANY_WAVES : HPRT_CLOSEST_WAVES) : 1024 / HPRT_TRACE_BLOCK) void k_trace(DevScene sc, const uint32_t *queue, const uint32_t *countPtr,
                                                            uint32_t countImm, RayStream rays, HitStream hits, uint8_t *occ,
                                                            DevCounters *counters, uint4 *rayStats, uint32_t *workCounter, uint32_t chunk,
                                                            TraceTune tune) {
    constexpr bool COUNT = MODE == 1, PROF = MODE == 2;
    constexpr bool PACKED = !QUAD;      // both children's slab test as packed two-float operations (the kernels with the quadric code keep the scalar form: no gain there)
    // An any-hit ray's answer does not depend on the order of the walk (its tMax never shrinks, every primitive test is
    // independent of the others), so the plain any-hit kernel visits children in storage order and skips the re-test of
    // popped entries: 8-11 % more rays per second than the reference's front-to-back order (bvh.cpp:381-388), which the
    // counting and profiling variants keep so that their node counts are the reference's.  Nearest-entry-first was
    // measured too: 2 % slower than storage order.  Closest-hit rays keep the reference's order: ties in t (shared edges)
    // are resolved by it.
    constexpr bool FREE_ORDER = ANY_HIT && MODE == 0;
    // DEFER (order-free any-hit walks of plain triangle scenes; -DHPRT_DEFER_LEAF, an experiment kept for the record): a lane that reaches a
    // leaf does not wait for the primitive phase with it but keeps it aside (`pend`, one entry) and walks on with its stack, so that the phase
    // finds more lanes with a leaf.  Bit-identical films; any-hit rays per second -2.8 % on the atrium, -2.0 % in the living
    // room (profiles/r03_defer_leaf_ab.txt): the steps a lane takes before its leaf is tested are wasted whenever that leaf holds the hit.
#ifdef HPRT_DEFER_LEAF
    constexpr bool DEFER = FREE_ORDER && !INST && !QUAD;
#else
    constexpr bool DEFER = false;
#endif
    int pend = REF_NONE;
    constexpr bool HPRT_INLINE_PRETEST = true;
    unsigned long long pf[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned int pfPush = 0u, pfSpill = 0u;      // PROF: stack pushes, and those beyond the LDS entries (scratch)
    const unsigned long long pfStart = PROF ? clock64() : 0ull;
    // (instanced scenes: the world-space ray takes another 6 KB of LDS per workgroup, so their stacks keep fewer entries for the same occupancy)
    constexpr int LDS_N = MODE != 0 ? HPRT_LDS_STACK
                        : !INST ? (QUAD ? (ANY_HIT ? HPRT_LDS_STACK_QUAD_ANY : HPRT_LDS_STACK) : ANY_HIT ? HPRT_LDS_STACK_ANY : HPRT_LDS_STACK_CLOSEST)
                                : (QUAD ? HPRT_LDS_STACK_QUAD_ANY : ANY_HIT ? HPRT_LDS_STACK_INST_ANY : HPRT_LDS_STACK_INST_CLOSEST);
    __shared__ uint2 stackMem[LDS_N * HPRT_TRACE_BLOCK];     // [entry][thread]: {ref, tMin}
    uint2 *const ldsStack = &stackMem[threadIdx.x];
    // INST: the world-space ray stays in LDS ([component][thread]) while the lane walks an instance with the transformed one
    __shared__ float worldRayMem[INST ? 6 * HPRT_TRACE_BLOCK : 1];
    float *const worldRay = &worldRayMem[INST ? threadIdx.x : 0];
    const uint32_t n = countPtr ? *countPtr : countImm;
    const uint32_t lane = __lane_id();
    // Pairs and primitives are fetched with buffer loads: a 32-bit per-lane byte offset against a
    // wave-uniform descriptor (no 64-bit address arithmetic) and — unlike plain loads, which hipcc
    // splits and sinks into the branches that consume each component — exactly four 16-byte
    // requests per pair and three per primitive, issued together.
    const auto pairRsrc = __builtin_amdgcn_make_buffer_rsrc((void *)sc.pairs, 0, (int)(sc.nPairs * 64u), 0x00020000);
    const auto triRsrc = __builtin_amdgcn_make_buffer_rsrc((void *)sc.tris, 0, (int)(sc.nPrims * 48u), 0x00020000);
    // INST: translation and entry of the instance a top-level primitive stands for (dev_scene.h, topEntry); a request past the table reads as zero
    const auto entryRsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(INST ? sc.topEntry : nullptr), 0, (int)(INST ? sc.nTopPrims * 16u : 0u), 0x00020000);
    const float robust = 1 + 2 * gamma_n(3);
    TraceCount cnt = {0u, 0u, 0u, 0u, 0u};
    unsigned int snapEntered = 0u, snapLeaf = 0u, snapPrim = 0u;     // counter values when the lane's current ray started (per-ray statistics)
    // per-lane ray state
    bool active = false, hit = false;
    uint32_t slot = 0;
    vec3 ro, invDir;
    float rayTMax = 0.f;
    RayShear shear; shear.k0 = shear.k1 = false; shear.Sx = shear.Sy = shear.Sz = 0.f;
    // dirIsNeg of the reference (accelerators/bvh.cpp:358), as lane masks for the bound selects and as bits (bit a = invDir[a] < 0)
    // for the near-child decision by split axis
    bool ngX = false, ngY = false, ngZ = false;
    uint32_t negMask = 0u;
    int sp = 0, cur = REF_NONE;
    int32_t prim = -1; float hb0 = 0.f, hb1 = 0.f, hb2 = 0.f;
    // A parked quadric waits for the batched slow phase (wait = 1; waitInfo = sphere index | bit 31 "last primitive
    // of its leaf").  Object instances (TransformedPrimitive, core/primitive.cpp:77-102) are entered in the primitive
    // phase itself, and left there when the walk pops the REF_EXIT sentinel: both cost about as much as a triangle test.
    uint32_t wait = 0u, waitInfo = 0u;
    int inst = -1, hitInst = -1;          // instance being walked / instance of the closest hit so far
    bool instHit = false;                 // a hit was recorded inside the instance being walked
    uint32_t instPrim = 0u;               // top-level ordered index of that instance's primitive | bit 31 "last of its leaf"
    float savedTMax = 0.f;                // world tMax at the moment the instance was entered
    // Stack entries beyond the LDS ones live in a per-scene HBM area, [entry][thread of the grid] (rare: no ray of the measured
    // scenes, BVH depth up to 26, ever had more than 16 pending siblings).  (Private arrays for them made hipcc address scratch
    // through flat pointers, and its gfx950 back end rejects the null checks of those casts in some instantiations.)
    // (volatile: keeps hipcc from folding the LDS and the HBM access into one access through a generic pointer)
    // (the address is formed where it is used: the pointer would otherwise hold two registers for the whole walk)
    auto deepSlot = [&](int entry) -> volatile unsigned long long * {
        return (volatile unsigned long long *)sc.deepStack + (size_t)(entry - LDS_N) * HPRT_DEEP_THREADS + (blockIdx.x * HPRT_TRACE_BLOCK + threadIdx.x);
    };
    bool moreWork = n > 0 && sc.nPairs > 0;

    // nodesToVisit[--toVisitOffset] + the tMax-dependent part of the bounds test (see above)
    auto pop = [&]() -> int {
        while (sp > 0) {
            --sp;
            uint2 e;
            if (sp < LDS_N) e = ldsStack[sp * HPRT_TRACE_BLOCK]; else { const unsigned long long w = *deepSlot(sp); e = make_uint2((uint32_t)w, (uint32_t)(w >> 32)); }
            if (INST && (int)e.x == REF_EXIT) { savedTMax = __uint_as_float(e.y); return REF_EXIT; }     // the instance's walk is over
            if (COUNT) ++cnt.fetched;
            // (an any-hit ray's tMax never shrinks: what was pushed with tMin < tMax still passes)
            if (FREE_ORDER || __uint_as_float(e.y) < rayTMax) { if (COUNT) { ++cnt.entered; if ((int)e.x < 0) ++cnt.leaf; } return (int)e.x; }
        }
        return REF_NONE;
    };

    // The wave draws rays from the global queue head in chunks (one atomic per `chunk` rays:
    // a single word retires only ~90 atomics per microsecond) and hands them to idle lanes
    // from its private range [localNext, localEnd).
    uint32_t localNext = 0u, localEnd = 0u;
    uint32_t window = 0u, windowBase = 0xffffffffu;
    if (sc.nPairs == 0 && n > 0) {
        // empty aggregate: every ray misses (accelerators/bvh.cpp:355)
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
            const uint32_t s = queue ? queue[i] : i;
            if (ANY_HIT) occ[s] = 0;
            else { hits.a[s] = make_float4(rays.a[s].w, __int_as_float(-1), 0.f, 0.f); if (hits.b) hits.b[s] = make_float2(0.f, __int_as_float(-1)); }
        }
    }
    while (true) {
        // ---- refill idle lanes from the queue ----
        if (moreWork) {
            const unsigned long long idle = __ballot(!active);
            if (idle != 0ull) {
                const unsigned long long pfT = PROF ? clock64() : 0ull;
                if (localNext >= localEnd) {
                    uint32_t base = 0u;
                    if (lane == 0) base = atomicAdd(workCounter, chunk);
                    base = __shfl(base, 0);
                    localNext = base < n ? base : n;
                    localEnd = (base + chunk < n) ? base + chunk : n;
                    if (localNext >= localEnd) moreWork = false;       // queue drained
                    // the window prefetched at the end of the exhausted chunk is empty; it must not pass for the head of this
                    // chunk when the two happen to be adjacent (same base)
                    windowBase = 0xffffffffu;
                }
                const uint32_t want = (uint32_t)__popcll(idle);
                const uint32_t base = localNext;
                localNext = (localNext + want < localEnd) ? localNext + want : localEnd;
                // Queue entries are taken from a per-wave window (lane l holds queue[windowBase + l]) that was
                // loaded during the previous refill, so a refill pays one memory round trip (the rays), not two.
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                uint32_t slotW = 0u;
                if (queue) {
                    if (windowBase != base) { window = (base + lane < localEnd) ? queue[base + lane] : 0u; windowBase = base; }   // new chunk
                    slotW = __shfl(window, (int)rank);
                    window = (localNext + lane < localEnd) ? queue[localNext + lane] : 0u;      // for the next refill; arrives with the rays below
                    windowBase = localNext;
                }
                if (!active) {
                    const uint32_t idx = base + rank;
                    if (idx < localEnd) {
                        slot = queue ? slotW : idx;
                        const float4 ra = rays.a[slot], rb = rays.b[slot];
                        ro = vec3(ra.x, ra.y, ra.z);
                        const vec3 rd(rb.x, rb.y, rb.z);
                        rayTMax = ra.w;
                        invDir = vec3(1 / rd.x, 1 / rd.y, 1 / rd.z);
                                                ngX = invDir.x < 0; ngY = invDir.y < 0; ngZ = invDir.z < 0;
                        if (PACKED) negMask = (ngX ? 1u : 0u) | (ngY ? 2u : 0u) | (ngZ ? 4u : 0u);
                        shear = ray_shear(rd, invDir);
                        sp = 0; cur = 0; hit = false; prim = -1; hb0 = hb1 = hb2 = 0.f;
                        wait = 0u; inst = -1; hitInst = -1; instHit = false; pend = REF_NONE;
                        if (INST) {
                            worldRay[0] = ra.x; worldRay[HPRT_TRACE_BLOCK] = ra.y; worldRay[2 * HPRT_TRACE_BLOCK] = ra.z;
                            worldRay[3 * HPRT_TRACE_BLOCK] = rb.x; worldRay[4 * HPRT_TRACE_BLOCK] = rb.y; worldRay[5 * HPRT_TRACE_BLOCK] = rb.z;
                        }
                        if (COUNT) { snapEntered = cnt.entered; snapLeaf = cnt.leaf; snapPrim = cnt.tri + cnt.sphere; }
                        active = true;
                    }
                }
                if (PROF) { pf[1] += clock64() - pfT; pf[9] += 1; pf[10] += __popcll(idle); }
            }
        }
        if (__ballot(active) == 0ull) break;
        // ---- walk until too few lanes are busy (or, with the queue drained, until all are done) ----
        while (true) {
            // phase 1: wave-uniform loop of pair steps.  It ends early once `parkLimit` lanes hold a
            // parked primitive (or after `stepLimit` steps) so that parked lanes do not idle behind
            // the longest interior run.
            int steps = 0;
            const unsigned long long pfT1 = PROF ? clock64() : 0ull;
            while (true) {
                const bool trav = active && cur >= 0;
                if (__ballot(trav) == 0ull) break;
                if (PROF) { pf[5] += 1; pf[6] += __popcll(__ballot(trav)); }
                if (trav) {
                    // One pair = four 16-byte words: the x, y and z planes {lo0, lo1, hi0, hi1} and {ref0, ref1, meta, -}: four requests,
                    // issued together.  (Fetching the bounds in near/far order with six 8-byte requests at sign-dependent offsets
                    // saves the twelve selects below and was measured 15-20 % SLOWER: the step is bound by the texture-address /
                    // L1 pipeline as much as by the VALU, and that pipeline works per request, not per byte.)
                    const uint32_t base = (uint32_t)cur * 64u;
                    const u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(pairRsrc, base, 0, 0);
                    const u32x4 q1 = __builtin_amdgcn_raw_buffer_load_b128(pairRsrc, base + 16u, 0, 0);
                    const u32x4 q2 = __builtin_amdgcn_raw_buffer_load_b128(pairRsrc, base + 32u, 0, 0);
                    const u32x4 q3 = __builtin_amdgcn_raw_buffer_load_b128(pairRsrc, base + 48u, 0, 0);
                    // dirIsNeg picks, per axis, the bounds that give tMin ("near") and tMax ("far"), as {child 0, child 1}
                    float t0, t1;
                    bool s0, s1;
                    if (!PACKED) {
                        s0 = slab_test(__uint_as_float(q0.x), __uint_as_float(q0.z), __uint_as_float(q1.x), __uint_as_float(q1.z),
                                       __uint_as_float(q2.x), __uint_as_float(q2.z), ro, invDir, ngX, ngY, ngZ, robust, &t0);
                        s1 = slab_test(__uint_as_float(q0.y), __uint_as_float(q0.w), __uint_as_float(q1.y), __uint_as_float(q1.w),
                                       __uint_as_float(q2.y), __uint_as_float(q2.w), ro, invDir, ngX, ngY, ngZ, robust, &t1);
                    } else {
                        u32x2 rnx, rfx, rny, rfy, rnz, rfz;
                        rnx.x = ngX ? q0.z : q0.x; rnx.y = ngX ? q0.w : q0.y; rfx.x = ngX ? q0.x : q0.z; rfx.y = ngX ? q0.y : q0.w;
                        rny.x = ngY ? q1.z : q1.x; rny.y = ngY ? q1.w : q1.y; rfy.x = ngY ? q1.x : q1.z; rfy.y = ngY ? q1.y : q1.w;
                        rnz.x = ngZ ? q2.z : q2.x; rnz.y = ngZ ? q2.w : q2.y; rfz.x = ngZ ? q2.x : q2.z; rfz.y = ngZ ? q2.y : q2.w;
                        slab_test_pair(__builtin_bit_cast(f32x2, rnx), __builtin_bit_cast(f32x2, rfx), __builtin_bit_cast(f32x2, rny), __builtin_bit_cast(f32x2, rfy),
                                       __builtin_bit_cast(f32x2, rnz), __builtin_bit_cast(f32x2, rfz), ro.x, ro.y, ro.z, invDir.x, invDir.y, invDir.z, robust, &s0, &s1, &t0, &t1);
                    }
                    const uint32_t meta = q3.z;
                    const uint32_t axis = meta & 3u;
                    const bool single = (meta & PAIR_SINGLE) != 0u;
                    // second child first when the ray is negative along the split axis (bvh.cpp:381-388)
                    const bool isNeg = FREE_ORDER ? false : !PACKED ? (axis == 0 ? ngX : (axis == 1 ? ngY : ngZ)) : ((negMask >> axis) & 1u) != 0u;
                    const int refN = (int)(isNeg ? q3.y : q3.x), refF = (int)(isNeg ? q3.x : q3.y);
                    const float tN = isNeg ? t1 : t0, tF = isNeg ? t0 : t1;
                    const bool hitN = (isNeg ? s1 : s0) && tN < rayTMax;
                    const bool slabF = (isNeg ? s0 : s1) && !single;
                    if (COUNT) { ++cnt.fetched; if (hitN) { ++cnt.entered; if (refN < 0) ++cnt.leaf; } }
                    if (hitN) {
                        cur = refN;
                        if (COUNT ? !single : (slabF && tF < rayTMax)) {
                            const uint2 e = make_uint2((uint32_t)refF, __float_as_uint(slabF ? tF : HPRT_INF));
                            if (PROF) { ++pfPush; if (sp >= LDS_N) ++pfSpill; }
                            if (sp < LDS_N) { ldsStack[sp * HPRT_TRACE_BLOCK] = e; ++sp; }
                            else if (sp < HPRT_STACK_TOTAL) { *deepSlot(sp) = (unsigned long long)e.x | ((unsigned long long)e.y << 32); ++sp; }
                        }
                    } else {
                        const bool hitF = slabF && tF < rayTMax;
                        if (COUNT && !single) { ++cnt.fetched; if (hitF) { ++cnt.entered; if (refF < 0) ++cnt.leaf; } }
                        cur = hitF ? refF : pop();
                    }
                    if (DEFER) {
                        if (is_parked(cur) && pend == REF_NONE && sp > 0) { pend = cur; cur = pop(); }      // set the leaf aside, walk on
                        else if (cur == REF_NONE && pend != REF_NONE) { cur = pend; pend = REF_NONE; }      // nothing left to walk: the leaf set aside
                    }
                }
                ++steps;
                if (steps >= tune.stepLimit || __popcll(__ballot(active && (INST ? wants_prim_phase(cur) : is_parked(cur)) && wait == 0u)) >= tune.parkLimit) break;
            }
            const unsigned long long pfT2 = PROF ? clock64() : 0ull;
            if (PROF) pf[2] += pfT2 - pfT1;
            unsigned long long pfSphere = 0ull;
            // phase 2: the parked primitives (wave-uniform loop over the longest leaf run).
            // Quadric primitives are expensive (interval arithmetic) and reached by lanes at
            // different times, so a lane that meets one waits (wait = 1) until `sphereLimit`
            // lanes wait or nothing else can run, and the test runs once for all of them.
            while (true) {
                const bool hasPend = DEFER && active && pend != REF_NONE && cur >= 0;      // walking, with a leaf set aside
                const bool todo = active && ((INST ? wants_prim_phase(cur) : is_parked(cur)) || hasPend) && wait == 0u;
                const int nPending = __popcll(__ballot(todo));
                // too few parked lanes for a primitive test to pay: let the others walk first
                if (nPending != 0 && nPending < tune.primMin && __ballot(active && cur >= 0) != 0ull) break;
                if (nPending != 0) {
                    if (hasPend) { const int t = cur; cur = pend; pend = t; }      // test the leaf now; the pair to walk on with waits in `pend`
                    if (PROF) { pf[7] += 1; pf[8] += nPending; }
                    // (two-level scenes: leaving an instance, entering one and a triangle test share the iteration.  One kind per iteration — the
                    // one most lanes wait for — was measured: twice the iterations at 55 % of the cycles each, 13 % fewer rays per second on
                    // instanced-10m.  An iteration costs its memory round trip, not its instructions.)
                    // (Measured alternatives for leaving an instance, instanced-10m, closest / any-hit rays per second against this form: the world
                    // ray's 1 / d and shear kept in LDS instead of recomputed (six divisions): -5 % / -5 %; the same at the head of the
                    // pair steps: -10 % / -6 %; inside pop(): -51 % / -46 %.)
                    if (INST && todo && cur == REF_EXIT) {
                        // the instance's walk is over, back to world space: r.tMax = ray.tMax only if the instance was
                        // hit (core/primitive.cpp:85-86); continue with the top-level leaf the instance belongs to
                        const float worldT = instHit ? rayTMax : savedTMax;
                        ro = vec3(worldRay[0], worldRay[HPRT_TRACE_BLOCK], worldRay[2 * HPRT_TRACE_BLOCK]);
                        const vec3 rd(worldRay[3 * HPRT_TRACE_BLOCK], worldRay[4 * HPRT_TRACE_BLOCK], worldRay[5 * HPRT_TRACE_BLOCK]);
                        invDir = vec3(1 / rd.x, 1 / rd.y, 1 / rd.z);
                                                ngX = invDir.x < 0; ngY = invDir.y < 0; ngZ = invDir.z < 0;
                        if (PACKED) negMask = (ngX ? 1u : 0u) | (ngY ? 2u : 0u) | (ngZ ? 4u : 0u);
                        shear = ray_shear(rd, invDir);
                        rayTMax = worldT;
                        inst = -1; instHit = false;
                        if (instPrim & 0x80000000u) cur = pop();
                        else cur = ~(int)((instPrim & 0x7fffffffu) + 1u);
                    }
                    const bool pending = todo && is_parked(cur);      // (a lane that just left an instance may be at its next primitive)
                    if (pending) {
                        const uint32_t pi = (uint32_t)~cur;
                        u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(triRsrc, pi * 48, 0, 0);
                        u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(triRsrc, pi * 48 + 16, 0, 0);
                        u32x4 v2 = __builtin_amdgcn_raw_buffer_load_b128(triRsrc, pi * 48 + 32, 0, 0);
                        // keep the three requests whole and in flight together (hipcc otherwise narrows the
                        // second one and sinks it behind the tag test: a second, dependent memory round trip)
                        u32x4 ve = {0u, 0u, 0u, 0u};
                        if (INST) {
                            // a top-level primitive may be an instance: its translation and entry travel with the record (one round trip
                            // for the whole entry instead of record -> DevInstance)
                            if (inst < 0) ve = __builtin_amdgcn_raw_buffer_load_b128(entryRsrc, pi * 16, 0, 0);
                            asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(ve));
                        } else
                        asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2));
                        const uint32_t tag = v0.w;
                        if ((tag & TAG_KIND_MASK) == 0u) {
                            if (COUNT) ++cnt.tri;
                            float b0, b1, b2, t;
                            bool done = false;
                            if (tri_test(vec3(__uint_as_float(v0.x), __uint_as_float(v0.y), __uint_as_float(v0.z)),
                                         vec3(__uint_as_float(v1.x), __uint_as_float(v1.y), __uint_as_float(v1.z)),
                                         vec3(__uint_as_float(v2.x), __uint_as_float(v2.y), __uint_as_float(v2.z)), ro, rayTMax, shear, &b0, &b1, &b2, &t)) {
                                if (ANY_HIT) { hit = true; done = true; }
                                else if (!(tag & TAG_BOGUS)) { hit = true; rayTMax = t; prim = (int32_t)(pi | ((tag & TAG_BIN_MASK) << 24)); hb0 = b0; hb1 = b1; hb2 = b2; if (INST) { hitInst = inst; instHit = inst >= 0; } }
                            }
                            if (done) { cur = REF_NONE; pend = REF_NONE; }
                            else if (tag & TAG_LAST) { if (DEFER && pend != REF_NONE) { cur = pend; pend = REF_NONE; } else cur = pop(); }
                            else --cur;                                  // ~(pi + 1)
                        } else if (INST && (tag & TAG_KIND_MASK) == TAG_INSTANCE) {
                            // TransformedPrimitive::Intersect: Ray ray = Inverse(InterpolatedPrimToWorld)(r), i.e.
                            // Transform::operator()(const Ray &) (core/transform.h:251-264); then walk the object's aggregate
                            inst = (int)v2.w;
                            instPrim = pi | ((tag & TAG_LAST) ? 0x80000000u : 0u);
                            instHit = false;
                            mat4 W;
                            int root;
                            if (tag & TAG_INST_INLINE) {      // affine: the matrix came with the primitive
                                W.m[0][0] = __uint_as_float(v0.x); W.m[0][1] = __uint_as_float(v0.y); W.m[0][2] = __uint_as_float(v0.z); W.m[0][3] = __uint_as_float(ve.x);
                                W.m[1][0] = __uint_as_float(v1.x); W.m[1][1] = __uint_as_float(v1.y); W.m[1][2] = __uint_as_float(v1.z); W.m[1][3] = __uint_as_float(ve.y);
                                W.m[2][0] = __uint_as_float(v2.x); W.m[2][1] = __uint_as_float(v2.y); W.m[2][2] = __uint_as_float(v2.z); W.m[2][3] = __uint_as_float(ve.z);
                                W.m[3][0] = 0.f; W.m[3][1] = 0.f; W.m[3][2] = 0.f; W.m[3][3] = 1.f;
                                root = (int)ve.w;
                            } else { const DevInstance &in = sc.instances[inst]; W = in.w2i; root = in.root; }
                            vec3 oErr;
                            vec3 o2 = xf_point_err(W, ro, &oErr);
                            const vec3 d2 = xf_vector(W, vec3(worldRay[3 * HPRT_TRACE_BLOCK], worldRay[4 * HPRT_TRACE_BLOCK], worldRay[5 * HPRT_TRACE_BLOCK]));
                            const float lengthSquared = d2.x * d2.x + d2.y * d2.y + d2.z * d2.z;
                            float tm = rayTMax;
                            if (lengthSquared > 0) {
                                const float dt = dot(vabs(d2), oErr) / lengthSquared;
                                o2 = o2 + d2 * dt;
                                tm -= dt;
                            }
                            const uint2 e = make_uint2((uint32_t)REF_EXIT, __float_as_uint(rayTMax));
                            if (sp < LDS_N) ldsStack[sp * HPRT_TRACE_BLOCK] = e;
                            else if (sp < HPRT_STACK_TOTAL) *deepSlot(sp) = (unsigned long long)e.x | ((unsigned long long)e.y << 32);
                            ++sp;       // a full stack cannot take the sentinel: such depths are outside what the reference supports either (64 entries)
                            ro = o2; rayTMax = tm;
                            invDir = vec3(1 / d2.x, 1 / d2.y, 1 / d2.z);
                                                    ngX = invDir.x < 0; ngY = invDir.y < 0; ngZ = invDir.z < 0;
                        if (PACKED) negMask = (ngX ? 1u : 0u) | (ngY ? 2u : 0u) | (ngZ ? 4u : 0u);
                            shear = ray_shear(d2, invDir);
                            cur = root;
                        } else if (QUAD) {
                            // a quadric: the cheap exact pre-test (dev_intersect.h) settles most of them here; the rest wait for
                            // the batched interval-arithmetic test
                            bool maybe = true;
                            if (HPRT_INLINE_PRETEST) {
                                DRay rr; rr.o = ro; rr.tMax = rayTMax;
                                if (INST) rr.d = vec3(worldRay[3 * HPRT_TRACE_BLOCK], worldRay[4 * HPRT_TRACE_BLOCK], worldRay[5 * HPRT_TRACE_BLOCK]);
                                else { const float4 rb = rays.b[slot]; rr.d = vec3(rb.x, rb.y, rb.z); }
                                if (INST && inst >= 0) rr.d = xf_vector(sc.instances[inst].w2i, rr.d);
                                maybe = sphere_may_hit(sc.spheres[v2.w], rr);
                            }
                            if (maybe) { wait = 1u; waitInfo = v2.w | ((tag & TAG_LAST) ? 0x80000000u : 0u) | ((tag & TAG_BIN_MASK) == (BIN_TEXTURED << TAG_BIN_SHIFT) ? 0x40000000u : 0u); }
                            else { if (COUNT) ++cnt.sphere; if (tag & TAG_LAST) cur = pop(); else --cur; }
                        }
                    }
                    continue;
                }
                if (!QUAD) break;
                const bool slow = active && wait != 0u;
                const int nWait = __popcll(__ballot(slow));
                if (nWait == 0) break;
                const bool canWalk = __ballot(active && cur >= 0) != 0ull;
                if (nWait < tune.sphereLimit && canWalk) break;      // keep waiting, let the others walk
                const unsigned long long pfT3 = PROF ? clock64() : 0ull;
                if (PROF) { pf[11] += 1; pf[12] += nWait; }
                if (wait == 1u) {
                    const uint32_t pi = (uint32_t)~cur;
                    wait = 0u;
                    if (COUNT) ++cnt.sphere;
                    DRay rr; rr.o = ro; rr.tMax = rayTMax;
                    if (INST) rr.d = vec3(worldRay[3 * HPRT_TRACE_BLOCK], worldRay[4 * HPRT_TRACE_BLOCK], worldRay[5 * HPRT_TRACE_BLOCK]);
                    else { const float4 rb = rays.b[slot]; rr.d = vec3(rb.x, rb.y, rb.z); }
                    if (INST && inst >= 0) rr.d = xf_vector(sc.instances[inst].w2i, rr.d);      // the instance-space direction, recomputed
                    DRay robj; vec3 ph; float phi, t;
                    bool done = false;
                    // the cheap exact pre-test (dev_intersect.h) settles most quadrics; the interval arithmetic runs for the rest
                    if ((HPRT_INLINE_PRETEST || sphere_may_hit(sc.spheres[waitInfo & 0x3fffffffu], rr)) && sphere_test(sc.spheres[waitInfo & 0x3fffffffu], rr, &robj, &ph, &phi, &t)) {
                        if (ANY_HIT) { hit = true; done = true; }
                        else { hit = true; rayTMax = t; prim = (int32_t)(pi | (((waitInfo & 0x40000000u) ? BIN_TEXTURED : BIN_GENERIC) << HIT_BIN_SHIFT)); hb0 = 0.f; hb1 = 0.f; hb2 = 0.f; if (INST) { hitInst = inst; instHit = inst >= 0; } }
                    }
                    if (done) cur = REF_NONE;
                    else if (waitInfo & 0x80000000u) cur = pop();
                    else --cur;
                }
                if (PROF) pfSphere += clock64() - pfT3;
            }
            if (PROF) { pf[3] += clock64() - pfT2 - pfSphere; pf[4] += pfSphere; }
            // retire finished rays
            if (active && cur == REF_NONE) {
                // per-ray Ray::stats of the fork (core/geometry.h:1078-1173): interior nodes entered, leaves entered, primitive tests
                if (COUNT && rayStats) rayStats[slot] = make_uint4(cnt.entered - snapEntered - (cnt.leaf - snapLeaf), cnt.leaf - snapLeaf, cnt.tri + cnt.sphere - snapPrim, 0u);
                if (ANY_HIT) occ[slot] = hit ? 1 : 0;
                else {
                    hits.a[slot] = make_float4(rayTMax, __int_as_float(hit ? prim : -1), hb0, hb1);
                    if (hits.b) hits.b[slot] = make_float2(hb2, __int_as_float(INST && hit ? hitInst : -1));
                }
                active = false;
            }
            const int busy = __popcll(__ballot(active));
            if (busy == 0) break;
            if (moreWork && busy < tune.refillBelow) break;
        }
    }
    if (COUNT) wave_count_add(counters, ANY_HIT, cnt);
    if (PROF) {
        for (int off = 32; off > 0; off >>= 1) { pfPush += __shfl_down(pfPush, off); pfSpill += __shfl_down(pfSpill, off); }
        if (lane == 0) { atomicAdd(&g_traceProf[(ANY_HIT ? 16 : 0) + 14], (unsigned long long)pfPush); atomicAdd(&g_traceProf[(ANY_HIT ? 16 : 0) + 15], (unsigned long long)pfSpill); }
    }
    if (PROF && lane == 0) {
        pf[0] = clock64() - pfStart;
        for (int k = 0; k < 13; ++k) atomicAdd(&g_traceProf[(ANY_HIT ? 16 : 0) + k], pf[k]);
        atomicAdd(&g_traceProf[(ANY_HIT ? 16 : 0) + 13], 1ull);
    }
}

// ---------------------------------------------------------------------------
// k_walk4: the leaf-exact walk of plain renders (wide_bvh.h, dev_wide.h) — the same persistent waves, the same two
// wave-uniform phases and the same primitive tests as k_trace, over four-wide records with quantised child boxes.
//
// Why it returns what BVHAccel::Intersect / IntersectP return (accelerators/bvh.cpp:354-437):
// Bounds3::IntersectP(ray, invDir, dirIsNeg) (core/geometry.h:1754-1780) is monotone under box inclusion and a node's box
// contains its descendants' boxes exactly (Union of primitive bounds, bvh.cpp:220-222); tMax only shrinks.  So the
// reference's walk tests the primitives of a leaf if and only if that leaf's OWN box passes against the tMax current when
// the leaf comes up, and the leaves come up in an order fixed by the tree and the ray's signs.  Here the leaves come up in
// that same order (slots visited by the nested near / far decisions of the three collapsed split axes; any-hit rays, whose
// answer is order-free, in storage order), every leaf's own exact box is tested with the reference's arithmetic
// (slab_test) before its primitives count — for a one-triangle leaf from the vertices just fetched, otherwise from
// DevScene::leafBox — and the interior tests are conservative: a dequantised box contains the exact one, the same float
// operations run on it, NaN comparisons never cull, so whatever the exact test passes, this one passes.  What is culled
// coarser costs work, never a result.  An any-hit ray tests the leaf's exact box only when a triangle reports a hit (the
// hit counts if the box passes, else the leaf is one the reference never reaches and is left).
// Counting and profiling renders keep k_trace: their node counts are the reference's.
// ---------------------------------------------------------------------------
#ifndef HPRT_WALK4_CLOSEST_WAVES
#define HPRT_WALK4_CLOSEST_WAVES 6
#endif
#ifndef HPRT_WALK4_ANY_WAVES
#define HPRT_WALK4_ANY_WAVES 6
#endif
#ifndef HPRT_WALK4_INST_CLOSEST_WAVES
#define HPRT_WALK4_INST_CLOSEST_WAVES 5
#endif
#ifndef HPRT_WALK4_INST_ANY_WAVES
#define HPRT_WALK4_INST_ANY_WAVES 6
#endif
// Entry and exit distances of the four dequantised slot boxes of a wide record: the operations of Bounds3::IntersectP, packed so that the
// ray's operands are pairs it holds anyway — x and y of one slot share an instruction ({x, y} of the origin, the reciprocal direction, the
// grid), z takes two slots at a time.  (A first version paired {slot, slot}: hipcc then kept every ray operand twice, {v, v}: six registers for nothing.)
__device__ __forceinline__ void wide_slab4(uint32_t nX, uint32_t fX, uint32_t nY, uint32_t fY, uint32_t nZ, uint32_t fZ, f32x2 sxy, float sz, f32x2 oxy, float oz,
                                           f32x2 roxy, float roz, f32x2 ivxy, float ivz, float robust, float tE[4], float tX[4]) {
    f32x2 tn[4], tf[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x2 qn = {(float)((nX >> (8 * k)) & 0xffu), (float)((nY >> (8 * k)) & 0xffu)}, qf = {(float)((fX >> (8 * k)) & 0xffu), (float)((fY >> (8 * k)) & 0xffu)};
        const f32x2 bn = __builtin_elementwise_fma(qn, sxy, oxy), bf = __builtin_elementwise_fma(qf, sxy, oxy);
        tn[k] = (bn - roxy) * ivxy;
        tf[k] = ((bf - roxy) * ivxy) * robust;
    }
    const f32x2 vsz = {sz, sz}, voz = {oz, oz};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2 qn = {(float)((nZ >> (16 * h)) & 0xffu), (float)((nZ >> (16 * h + 8)) & 0xffu)}, qf = {(float)((fZ >> (16 * h)) & 0xffu), (float)((fZ >> (16 * h + 8)) & 0xffu)};
        const f32x2 bn = __builtin_elementwise_fma(qn, vsz, voz), bf = __builtin_elementwise_fma(qf, vsz, voz);
        const f32x2 tnz = (bn - roz) * ivz;
        const f32x2 tfz = ((bf - roz) * ivz) * robust;
        tE[2 * h] = fmaxf(fmaxf(tn[2 * h].x, tn[2 * h].y), tnz.x); tE[2 * h + 1] = fmaxf(fmaxf(tn[2 * h + 1].x, tn[2 * h + 1].y), tnz.y);
        tX[2 * h] = fminf(fminf(tf[2 * h].x, tf[2 * h].y), tfz.x); tX[2 * h + 1] = fminf(fminf(tf[2 * h + 1].x, tf[2 * h + 1].y), tfz.y);
    }
}

// PROF (HPRT_TRACE_PROFILE=1, tools/trace_profile.py): phase cycles and lane counts into g_traceProf, as k_trace<., 2>; [11] leaves whose
// box was tested, [12] leaves whose box passed (closest hit) or hits whose leaf was checked / confirmed (any hit).
// INST: object instances (TransformedPrimitive::Intersect / IntersectP, core/primitive.cpp:77-102) as in k_trace — a lane that meets an
// instance primitive transforms its ray, pushes the REF_EXIT sentinel and walks the object's own wide records on the same stack; the
// world tMax of the moment waits in a register (instances do not nest).  The leaf that holds the instance is a leaf like any other: its
// exact box (the instance's world bound) decides whether the reference's walk reaches it.
// QUAD: quadrics, with k_trace's exact pre-test and batched interval-arithmetic test.
template <bool ANY_HIT, bool PROF, bool INST, bool QUAD>
__global__ __launch_bounds__(HPRT_TRACE_BLOCK, PROF ? 4 : QUAD ? (ANY_HIT ? HPRT_QUAD_ANY_WAVES : HPRT_QUAD_CLOSEST_WAVES) : INST ? (ANY_HIT ? HPRT_WALK4_INST_ANY_WAVES : HPRT_WALK4_INST_CLOSEST_WAVES) : ANY_HIT ? HPRT_WALK4_ANY_WAVES : HPRT_WALK4_CLOSEST_WAVES)
void k_walk4(DevScene sc, const uint32_t *queue, const uint32_t *countPtr, uint32_t countImm, RayStream rays, HitStream hits, uint8_t *occ, uint32_t *workCounter, uint32_t chunk, TraceTune tune) {
    constexpr int LDS_N = ANY_HIT ? HPRT_WIDE_LDS_ANY : HPRT_WIDE_LDS_CLOSEST;
    // [entry][thread]; closest hit: {ref, entry distance of the dequantised box}; any hit: the reference alone
    __shared__ uint32_t stackMem[(ANY_HIT ? 1 : 2) * LDS_N * HPRT_TRACE_BLOCK];
    uint32_t *const ldsRef = &stackMem[threadIdx.x];
    uint32_t *const ldsT = &stackMem[(ANY_HIT ? 0 : LDS_N * HPRT_TRACE_BLOCK) + threadIdx.x];
    // INST: the world-space ray stays in LDS ([component][thread]) while the lane walks an instance with the transformed one
    __shared__ float worldRayMem[INST ? 6 * HPRT_TRACE_BLOCK : 1];
    float *const worldRay = &worldRayMem[INST ? threadIdx.x : 0];
    const uint32_t n = countPtr ? *countPtr : countImm;
    const uint32_t lane = __lane_id();
    unsigned long long pf[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned int pfPush = 0u, pfSpill = 0u, pfLeaf = 0u, pfLeafOk = 0u;
    const unsigned long long pfStart = PROF ? clock64() : 0ull;
    const auto wideRsrc = __builtin_amdgcn_make_buffer_rsrc((void *)sc.wide, 0, (int)(sc.nWide * 64u), 0x00020000);
    const auto triRsrc = __builtin_amdgcn_make_buffer_rsrc((void *)sc.tris, 0, (int)(sc.nPrims * 48u), 0x00020000);
    const auto boxRsrc = __builtin_amdgcn_make_buffer_rsrc((void *)sc.leafBox, 0, (int)(sc.nPrims * 32u), 0x00020000);
    // INST: translation and wide entry of the instance a top-level primitive stands for (dev_scene.h, topEntryWide)
    const auto entryRsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(INST ? sc.topEntryWide : nullptr), 0, (int)(INST ? sc.nTopPrims * 16u : 0u), 0x00020000);
    const float robust = 1 + 2 * gamma_n(3);
    bool active = false, hit = false;
    uint32_t slot = 0;
    vec3 ro, invDir;
    float rayTMax = 0.f;
    RayShear shear; shear.k0 = shear.k1 = false; shear.Sx = shear.Sy = shear.Sz = 0.f;
    bool ngX = false, ngY = false, ngZ = false;
    uint32_t negMask = 0u;
    int sp = 0, cur = REF_NONE;
    uint32_t wait = 0u, waitInfo = 0u;      // QUAD: a quadric waits for the batched test (k_trace)
    int inst = -1;                        // INST: instance being walked
    bool instHit = false;                 // a hit was recorded inside the instance being walked
    uint32_t instPrim = 0u;               // top-level ordered index of that instance's primitive | bit 31 "last of its leaf"
    float savedTMax = 0.f;                // world tMax at the moment the instance was entered
    auto deepSlot = [&](int entry) -> volatile unsigned long long * {
        return (volatile unsigned long long *)sc.deepStack + (size_t)(entry - LDS_N) * HPRT_DEEP_THREADS + (blockIdx.x * HPRT_TRACE_BLOCK + threadIdx.x);
    };
    auto push = [&](int ref, float t) {
        if (PROF) { ++pfPush; if (sp >= LDS_N) ++pfSpill; }
        if (sp < LDS_N) { ldsRef[sp * HPRT_TRACE_BLOCK] = (uint32_t)ref; if (!ANY_HIT) ldsT[sp * HPRT_TRACE_BLOCK] = __float_as_uint(t); ++sp; }
        else if (sp < HPRT_WIDE_STACK_MAX) { *deepSlot(sp) = (unsigned long long)(uint32_t)ref | ((unsigned long long)__float_as_uint(t) << 32); ++sp; }
    };
    // the next pending slot; for closest-hit rays only if it can still matter (its dequantised box is not entered behind the hit found since)
    auto pop = [&]() -> int {
        while (sp > 0) {
            --sp;
            uint32_t r, t;
            if (sp < LDS_N) { r = ldsRef[sp * HPRT_TRACE_BLOCK]; t = ANY_HIT ? 0u : ldsT[sp * HPRT_TRACE_BLOCK]; }
            else { const unsigned long long w = *deepSlot(sp); r = (uint32_t)w; t = (uint32_t)(w >> 32); }
            if (INST && (int)r == REF_EXIT) return REF_EXIT;      // the instance's walk is over
            if (ANY_HIT || !(__uint_as_float(t) >= rayTMax)) return (int)r;
        }
        return REF_NONE;
    };
    auto set_ray = [&](vec3 o, vec3 d) {
        ro = o;
        invDir = vec3(1 / d.x, 1 / d.y, 1 / d.z);
        ngX = invDir.x < 0; ngY = invDir.y < 0; ngZ = invDir.z < 0;
        negMask = (ngX ? 1u : 0u) | (ngY ? 2u : 0u) | (ngZ ? 4u : 0u);
        shear = ray_shear(d, invDir);
    };
    // the ray's direction in the space being walked (not kept in registers: the walk needs 1 / d and the shear, the quadric tests d)
    auto ray_dir = [&]() -> vec3 {
        vec3 d;
        if (INST) d = vec3(worldRay[3 * HPRT_TRACE_BLOCK], worldRay[4 * HPRT_TRACE_BLOCK], worldRay[5 * HPRT_TRACE_BLOCK]);
        else { const float4 rb = rays.b[slot]; d = vec3(rb.x, rb.y, rb.z); }
        if (INST && inst >= 0) d = xf_vector(sc.instances[inst].w2i, d);
        return d;
    };
    bool moreWork = n > 0;
    uint32_t localNext = 0u, localEnd = 0u;
    uint32_t window = 0u, windowBase = 0xffffffffu;
    while (true) {
        // ---- refill idle lanes from the queue (as k_trace) ----
        if (moreWork) {
            const unsigned long long idle = __ballot(!active);
            if (idle != 0ull) {
                const unsigned long long pfT = PROF ? clock64() : 0ull;
                if (localNext >= localEnd) {
                    uint32_t base = 0u;
                    if (lane == 0) base = atomicAdd(workCounter, chunk);
                    base = __builtin_amdgcn_readfirstlane(base);      // (wave-uniform from here on: the chunk bounds live in scalar registers)
                    localNext = base < n ? base : n;
                    localEnd = (base + chunk < n) ? base + chunk : n;
                    if (localNext >= localEnd) moreWork = false;
                    windowBase = 0xffffffffu;
                }
                const uint32_t want = (uint32_t)__popcll(idle);
                const uint32_t base = localNext;
                localNext = (localNext + want < localEnd) ? localNext + want : localEnd;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                uint32_t slotW = 0u;
                if (queue) {
                    if (windowBase != base) { window = (base + lane < localEnd) ? queue[base + lane] : 0u; windowBase = base; }
                    slotW = __shfl(window, (int)rank);
                    window = (localNext + lane < localEnd) ? queue[localNext + lane] : 0u;
                    windowBase = localNext;
                }
                if (!active) {
                    const uint32_t idx = base + rank;
                    if (idx < localEnd) {
                        slot = queue ? slotW : idx;
                        const float4 ra = rays.a[slot], rb = rays.b[slot];
                        rayTMax = ra.w;
                        set_ray(vec3(ra.x, ra.y, ra.z), vec3(rb.x, rb.y, rb.z));
                        sp = 0; cur = 0; hit = false; wait = 0u; inst = -1; instHit = false;
                        if (INST) {
                            worldRay[0] = ra.x; worldRay[HPRT_TRACE_BLOCK] = ra.y; worldRay[2 * HPRT_TRACE_BLOCK] = ra.z;
                            worldRay[3 * HPRT_TRACE_BLOCK] = rb.x; worldRay[4 * HPRT_TRACE_BLOCK] = rb.y; worldRay[5 * HPRT_TRACE_BLOCK] = rb.z;
                        }
                        active = true;
                    }
                }
                if (PROF) { pf[1] += clock64() - pfT; pf[9] += 1; pf[10] += __popcll(idle); }
            }
        }
        if (__ballot(active) == 0ull) break;
        while (true) {
            // phase 1: record steps
            int steps = 0;
            const unsigned long long pfT1 = PROF ? clock64() : 0ull;
            while (true) {
                const bool trav = active && cur >= 0;
                if (__ballot(trav) == 0ull) break;
                if (PROF) { pf[5] += 1; pf[6] += __popcll(__ballot(trav)); }
                if (trav) {
                    const uint32_t base = (uint32_t)cur * 64u;
                    const u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(wideRsrc, base, 0, 0);
                    const u32x4 q1 = __builtin_amdgcn_raw_buffer_load_b128(wideRsrc, base + 16u, 0, 0);
                    const u32x4 q2 = __builtin_amdgcn_raw_buffer_load_b128(wideRsrc, base + 32u, 0, 0);
                    const u32x4 q3 = __builtin_amdgcn_raw_buffer_load_b128(wideRsrc, base + 48u, 0, 0);
                    const uint32_t em = q0.w;
                    const float sx = __uint_as_float((em & 0xffu) << 23), sy = __uint_as_float((em & 0xff00u) << 15), sz = __uint_as_float((em & 0xff0000u) << 7);
                    // dirIsNeg picks, per axis, the bounds that give tMin ("near") and tMax ("far"): one select per word covers the four slots
                    const uint32_t nX = ngX ? q1.y : q1.x, fX = ngX ? q1.x : q1.y, nY = ngY ? q1.w : q1.z, fY = ngY ? q1.z : q1.w;
                    const uint32_t nZ = ngZ ? q2.y : q2.x, fZ = ngZ ? q2.x : q2.y;
                    float tE[4], tX[4];
                    wide_slab4(nX, fX, nY, fY, nZ, fZ, f32x2{sx, sy}, sz, f32x2{__uint_as_float(q0.x), __uint_as_float(q0.y)}, __uint_as_float(q0.z),
                               f32x2{ro.x, ro.y}, ro.z, f32x2{invDir.x, invDir.y}, invDir.z, robust, tE, tX);
                    int r[4] = {(int)q3.x, (int)q3.y, (int)q3.z, (int)q3.w};
                    // the reference's rejections, each as "not provably outside": a NaN (0 * inf on a grid plane) never culls.  (One comparison
                    // instead of three — max(tE, 0) > min(tX, tMax), a superset of what the reference passes — was measured 2-4 % slower.)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if ((tE[k] > tX[k]) || (tX[k] <= 0.f) || (tE[k] >= rayTMax)) r[k] = WIDE_NONE;
                    if (!ANY_HIT) {
                        // the order of the reference's walk: second child's slots first when the ray is negative along the record's split
                        // axis, and inside either child by that child's own axis (bvh.cpp:381-388, applied at both collapsed levels)
                        const uint32_t meta = em >> 24;
                        const bool n0 = ((negMask >> (meta & 3u)) & 1u) != 0u, nL = ((negMask >> ((meta >> 2) & 3u)) & 1u) != 0u, nR = ((negMask >> ((meta >> 4) & 3u)) & 1u) != 0u;
                        { const int a = nL ? r[1] : r[0], b = nL ? r[0] : r[1]; const float ta = nL ? tE[1] : tE[0], tb = nL ? tE[0] : tE[1]; r[0] = a; r[1] = b; tE[0] = ta; tE[1] = tb; }
                        { const int a = nR ? r[3] : r[2], b = nR ? r[2] : r[3]; const float ta = nR ? tE[3] : tE[2], tb = nR ? tE[2] : tE[3]; r[2] = a; r[3] = b; tE[2] = ta; tE[3] = tb; }
                        { const int a = n0 ? r[2] : r[0], b = n0 ? r[3] : r[1], c = n0 ? r[0] : r[2], d = n0 ? r[1] : r[3];
                          const float ta = n0 ? tE[2] : tE[0], tb = n0 ? tE[3] : tE[1], tc = n0 ? tE[0] : tE[2], td = n0 ? tE[1] : tE[3];
                          r[0] = a; r[1] = b; r[2] = c; r[3] = d; tE[0] = ta; tE[1] = tb; tE[2] = tc; tE[3] = td; }
                    }
                    const bool v0 = r[0] != WIDE_NONE, v1 = r[1] != WIDE_NONE, v2 = r[2] != WIDE_NONE, v3 = r[3] != WIDE_NONE;
                    // the first slot hit is walked now, the others wait, the next one on top
                    if (sp + 3 <= LDS_N) {
                        // the usual case: all three fit the LDS entries.  No branches: every candidate is written at the top, and the top only
                        // moves past the ones that count (what lies above the top is never read)
                        const int p3 = (v3 && (v0 || v1 || v2)) ? 1 : 0, p2 = (v2 && (v0 || v1)) ? 1 : 0, p1 = (v1 && v0) ? 1 : 0;
                        ldsRef[sp * HPRT_TRACE_BLOCK] = (uint32_t)r[3]; if (!ANY_HIT) ldsT[sp * HPRT_TRACE_BLOCK] = __float_as_uint(tE[3]); sp += p3;
                        ldsRef[sp * HPRT_TRACE_BLOCK] = (uint32_t)r[2]; if (!ANY_HIT) ldsT[sp * HPRT_TRACE_BLOCK] = __float_as_uint(tE[2]); sp += p2;
                        ldsRef[sp * HPRT_TRACE_BLOCK] = (uint32_t)r[1]; if (!ANY_HIT) ldsT[sp * HPRT_TRACE_BLOCK] = __float_as_uint(tE[1]); sp += p1;
                        if (PROF) pfPush += (unsigned)(p3 + p2 + p1);
                    } else {
                        if (v3 && (v0 || v1 || v2)) push(r[3], tE[3]);
                        if (v2 && (v0 || v1)) push(r[2], tE[2]);
                        if (v1 && v0) push(r[1], tE[1]);
                    }
                    cur = v0 ? r[0] : v1 ? r[1] : v2 ? r[2] : r[3];
                    if (cur == WIDE_NONE) cur = pop();
                }
                ++steps;
                if (steps >= tune.stepLimit || __popcll(__ballot(active && (INST ? wants_prim_phase(cur) : is_parked(cur)) && wait == 0u)) >= tune.parkLimit) break;
            }
            const unsigned long long pfT2 = PROF ? clock64() : 0ull;
            if (PROF) pf[2] += pfT2 - pfT1;
            unsigned long long pfSphere = 0ull;
            // phase 2: the parked leaves (and, INST, lanes that leave an instance)
            while (true) {
                const bool todo = active && (INST ? wants_prim_phase(cur) : is_parked(cur)) && wait == 0u;
                const int nPending = __popcll(__ballot(todo));
                if (nPending != 0 && nPending < tune.primMin && __ballot(active && cur >= 0) != 0ull) break;
                if (nPending != 0) {
                    if (PROF) { pf[7] += 1; pf[8] += nPending; }
                    if (INST && todo && cur == REF_EXIT) {
                        // the instance's walk is over, back to world space: r.tMax = ray.tMax only if the instance was hit
                        // (core/primitive.cpp:85-86); continue with the top-level leaf the instance belongs to
                        const float worldT = instHit ? rayTMax : savedTMax;
                        set_ray(vec3(worldRay[0], worldRay[HPRT_TRACE_BLOCK], worldRay[2 * HPRT_TRACE_BLOCK]),
                                vec3(worldRay[3 * HPRT_TRACE_BLOCK], worldRay[4 * HPRT_TRACE_BLOCK], worldRay[5 * HPRT_TRACE_BLOCK]));
                        rayTMax = worldT;
                        inst = -1; instHit = false;
                        if (instPrim & 0x80000000u) cur = pop();
                        else cur = (int)((~((instPrim & 0x7fffffffu) + 1u)) & ~WIDE_LEAF_FIRST);
                    }
                    const bool pending = todo && is_parked(cur);      // (a lane that just left an instance may be at its next primitive)
                    if (pending) {
                        const uint32_t c = (uint32_t)cur;
                        const uint32_t pi = ~(c | (WIDE_LEAF_BOXED | WIDE_LEAF_FIRST));
                        const bool entrySingle = (c & (WIDE_LEAF_BOXED | WIDE_LEAF_FIRST)) == (WIDE_LEAF_BOXED | WIDE_LEAF_FIRST), entryBoxed = (c & WIDE_LEAF_BOXED) == 0u;
                        u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(triRsrc, pi * 48, 0, 0);
                        u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(triRsrc, pi * 48 + 16, 0, 0);
                        u32x4 v2 = __builtin_amdgcn_raw_buffer_load_b128(triRsrc, pi * 48 + 32, 0, 0);
                        u32x4 x0 = {0u, 0u, 0u, 0u}, x1 = {0u, 0u, 0u, 0u}, ve = {0u, 0u, 0u, 0u};
                        // closest hit: the leaf's box decides before anything in it is tested; any hit: only a reported hit asks for it — except for
                        // quadrics and instances, whose tests are worth skipping (and an instance must not be entered at all if the walk never gets there)
                        if (entryBoxed && (!ANY_HIT || INST || QUAD)) { x0 = __builtin_amdgcn_raw_buffer_load_b128(boxRsrc, pi * 32, 0, 0); x1 = __builtin_amdgcn_raw_buffer_load_b128(boxRsrc, pi * 32 + 16, 0, 0); }
                        if (INST && inst < 0) ve = __builtin_amdgcn_raw_buffer_load_b128(entryRsrc, pi * 16, 0, 0);
                        if (INST) asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(x0), "+v"(x1), "+v"(ve));
                        else asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(x0), "+v"(x1));
                        const uint32_t tag = v0.w;
                        const vec3 p0(__uint_as_float(v0.x), __uint_as_float(v0.y), __uint_as_float(v0.z)), p1(__uint_as_float(v1.x), __uint_as_float(v1.y), __uint_as_float(v1.z)),
                                   p2(__uint_as_float(v2.x), __uint_as_float(v2.y), __uint_as_float(v2.z));
                        // the leaf's own box, exactly: Bounds3::IntersectP as the reference evaluates it for this node (slab_test + tMin < tMax)
                        auto leaf_reached = [&](bool fromVertices, u32x4 b0, u32x4 b1) -> bool {
                            float lx = __uint_as_float(b0.x), ly = __uint_as_float(b0.y), lz = __uint_as_float(b0.z), hx = __uint_as_float(b0.w), hy = __uint_as_float(b1.x), hz = __uint_as_float(b1.y);
                            if (fromVertices) {
                                lx = fminf(fminf(p0.x, p1.x), p2.x); ly = fminf(fminf(p0.y, p1.y), p2.y); lz = fminf(fminf(p0.z, p1.z), p2.z);
                                hx = fmaxf(fmaxf(p0.x, p1.x), p2.x); hy = fmaxf(fmaxf(p0.y, p1.y), p2.y); hz = fmaxf(fmaxf(p0.z, p1.z), p2.z);
                            }
                            float tEn;
                            return slab_test(lx, hx, ly, hy, lz, hz, ro, invDir, ngX, ngY, ngZ, robust, &tEn) && tEn < rayTMax;
                        };
                        const bool isTri = (tag & TAG_KIND_MASK) == 0u;
                        bool leave = false;      // done with this leaf: next pending slot
                        // the entry test of the leaf: always for closest-hit rays; for any-hit rays up front only where it saves an expensive test
                        bool reached = true;
                        if ((entrySingle || entryBoxed) && (!ANY_HIT || ((INST || QUAD) && entryBoxed && !isTri))) {
                            reached = leaf_reached(entrySingle, x0, x1);
                            if (PROF) { ++pfLeaf; if (reached) ++pfLeafOk; }
                        }
                        // any-hit rays: has the box of the leaf this primitive belongs to been tested (and passed)?
                        const bool boxKnown = !ANY_HIT || ((INST || QUAD) && entryBoxed && !isTri);
                        if (!reached) leave = true;
                        else if (isTri) {
                            float b0, b1, b2, t;
                            if (tri_test(p0, p1, p2, ro, rayTMax, shear, &b0, &b1, &b2, &t)) {
                                if (!ANY_HIT) {
                                    if (!(tag & TAG_BOGUS)) {
                                        // the record goes out now (a closer hit overwrites it): four registers fewer to carry through the walk, which is
                                        // what lets this kernel run six waves per SIMD; a ray that ends without a hit writes its miss record when it retires
                                        hit = true; rayTMax = t;
                                        hits.a[slot] = make_float4(t, __int_as_float((int32_t)(pi | ((tag & TAG_BIN_MASK) << 24))), b0, b1);
                                        if (hits.b) hits.b[slot] = make_float2(b2, __int_as_float(INST ? inst : -1));
                                        if (INST) instHit = inst >= 0;
                                    }
                                    if (tag & TAG_LAST) leave = true; else cur = (int)((~(pi + 1u)) & ~WIDE_LEAF_FIRST);
                                } else {
                                    // a hit counts if the reference's walk reaches this leaf; if it does not, none of the leaf's primitives is ever tested
                                    bool ok = true;
                                    if (c & WIDE_LEAF_FIRST) {      // (a later primitive of a leaf is only ever reached through a tested box: see below)
                                        if (!entrySingle) { x0 = __builtin_amdgcn_raw_buffer_load_b128(boxRsrc, pi * 32, 0, 0); x1 = __builtin_amdgcn_raw_buffer_load_b128(boxRsrc, pi * 32 + 16, 0, 0); }
                                        if (PROF) ++pfLeaf;
                                        ok = leaf_reached(entrySingle, x0, x1);
                                        if (PROF && ok) ++pfLeafOk;
                                    }
                                    if (ok) { hit = true; cur = REF_NONE; } else leave = true;
                                }
                            } else if (tag & TAG_LAST) leave = true;
                            else if (!ANY_HIT) cur = (int)((~(pi + 1u)) & ~WIDE_LEAF_FIRST);
                            else {
                                // any hit, a leaf of several primitives whose first one missed: the box has not been asked yet; the next primitive keeps
                                // the "entry" state (boxed), so that a hit on it still checks the leaf's box
                                cur = (int)((~(pi + 1u)) & ~WIDE_LEAF_BOXED);
                            }
                        } else if (INST && (tag & TAG_KIND_MASK) == TAG_INSTANCE) {
                            // TransformedPrimitive::Intersect: Ray ray = Inverse(InterpolatedPrimToWorld)(r), i.e.
                            // Transform::operator()(const Ray &) (core/transform.h:251-264); then walk the object's aggregate
                            inst = (int)v2.w;
                            instPrim = pi | ((tag & TAG_LAST) ? 0x80000000u : 0u);
                            instHit = false;
                            mat4 W;
                            int root;
                            if (tag & TAG_INST_INLINE) {      // affine: the matrix came with the primitive
                                W.m[0][0] = __uint_as_float(v0.x); W.m[0][1] = __uint_as_float(v0.y); W.m[0][2] = __uint_as_float(v0.z); W.m[0][3] = __uint_as_float(ve.x);
                                W.m[1][0] = __uint_as_float(v1.x); W.m[1][1] = __uint_as_float(v1.y); W.m[1][2] = __uint_as_float(v1.z); W.m[1][3] = __uint_as_float(ve.y);
                                W.m[2][0] = __uint_as_float(v2.x); W.m[2][1] = __uint_as_float(v2.y); W.m[2][2] = __uint_as_float(v2.z); W.m[2][3] = __uint_as_float(ve.z);
                                W.m[3][0] = 0.f; W.m[3][1] = 0.f; W.m[3][2] = 0.f; W.m[3][3] = 1.f;
                                root = (int)ve.w;
                            } else { const DevInstance &in = sc.instances[inst]; W = in.w2i; root = (int)in.pad[0]; }      // (pad[0]: the object's wide entry)
                            vec3 oErr;
                            vec3 o2 = xf_point_err(W, ro, &oErr);
                            const vec3 d2 = xf_vector(W, vec3(worldRay[3 * HPRT_TRACE_BLOCK], worldRay[4 * HPRT_TRACE_BLOCK], worldRay[5 * HPRT_TRACE_BLOCK]));
                            const float lengthSquared = d2.x * d2.x + d2.y * d2.y + d2.z * d2.z;
                            float tm = rayTMax;
                            if (lengthSquared > 0) {
                                const float dt = dot(vabs(d2), oErr) / lengthSquared;
                                o2 = o2 + d2 * dt;
                                tm -= dt;
                            }
                            savedTMax = rayTMax;
                            push(REF_EXIT, 0.f);
                            rayTMax = tm;
                            set_ray(o2, d2);
                            cur = root;
                        } else if (QUAD) {
                            // a quadric: the cheap exact pre-test (dev_intersect.h) settles most of them here; the rest wait for
                            // the batched interval-arithmetic test
                            DRay rr; rr.o = ro; rr.tMax = rayTMax; rr.d = ray_dir();
                            const bool maybe = sphere_may_hit(sc.spheres[v2.w], rr);
                            if (maybe) { wait = 1u; waitInfo = v2.w | ((tag & TAG_LAST) ? 0x80000000u : 0u) | ((tag & TAG_BIN_MASK) == (BIN_TEXTURED << TAG_BIN_SHIFT) ? 0x40000000u : 0u) | (boxKnown ? 0x20000000u : 0u); }
                            else if (tag & TAG_LAST) leave = true;
                            else cur = (int)((~(pi + 1u)) & ~(boxKnown ? WIDE_LEAF_FIRST : WIDE_LEAF_BOXED));
                        }
                        if (leave) cur = pop();
                    }
                    continue;
                }
                if (!QUAD) break;
                const bool slow = active && wait != 0u;
                const int nWait = __popcll(__ballot(slow));
                if (nWait == 0) break;
                const bool canWalk = __ballot(active && cur >= 0) != 0ull;
                if (nWait < tune.sphereLimit && canWalk) break;      // keep waiting, let the others walk
                const unsigned long long pfT3 = PROF ? clock64() : 0ull;
                if (wait == 1u) {
                    const uint32_t pi = ~((uint32_t)cur | (WIDE_LEAF_BOXED | WIDE_LEAF_FIRST));
                    wait = 0u;
                    DRay rr; rr.o = ro; rr.tMax = rayTMax; rr.d = ray_dir();
                    DRay robj; vec3 ph; float phi, t;
                    bool done = false, leave = false;
                    if (sphere_test(sc.spheres[waitInfo & 0x1fffffffu], rr, &robj, &ph, &phi, &t)) {
                        bool ok = true;
                        if (ANY_HIT && !(waitInfo & 0x20000000u)) {      // the leaf's box has not been asked yet
                            const u32x4 x0 = __builtin_amdgcn_raw_buffer_load_b128(boxRsrc, pi * 32, 0, 0), x1 = __builtin_amdgcn_raw_buffer_load_b128(boxRsrc, pi * 32 + 16, 0, 0);
                            float tEn;
                            ok = slab_test(__uint_as_float(x0.x), __uint_as_float(x0.w), __uint_as_float(x0.y), __uint_as_float(x1.x), __uint_as_float(x0.z), __uint_as_float(x1.y),
                                           ro, invDir, ngX, ngY, ngZ, robust, &tEn) && tEn < rayTMax;
                        }
                        if (!ok) leave = true;
                        else if (ANY_HIT) { hit = true; done = true; }
                        else {
                            hit = true; rayTMax = t;
                            hits.a[slot] = make_float4(t, __int_as_float((int32_t)(pi | (((waitInfo & 0x40000000u) ? BIN_TEXTURED : BIN_GENERIC) << HIT_BIN_SHIFT))), 0.f, 0.f);
                            if (hits.b) hits.b[slot] = make_float2(0.f, __int_as_float(INST ? inst : -1));
                            if (INST) instHit = inst >= 0;
                        }
                    }
                    if (done) cur = REF_NONE;
                    else if (leave || (waitInfo & 0x80000000u)) cur = pop();
                    else cur = (int)((~(pi + 1u)) & ~(((waitInfo & 0x20000000u) || !ANY_HIT) ? WIDE_LEAF_FIRST : WIDE_LEAF_BOXED));
                }
                if (PROF) pfSphere += clock64() - pfT3;
            }
            if (PROF) { pf[3] += clock64() - pfT2 - pfSphere; pf[4] += pfSphere; }
            // retire finished rays
            if (active && cur == REF_NONE) {
                if (ANY_HIT) occ[slot] = hit ? 1 : 0;
                else if (!hit) {
                    hits.a[slot] = make_float4(rayTMax, __int_as_float(-1), 0.f, 0.f);
                    if (hits.b) hits.b[slot] = make_float2(0.f, __int_as_float(-1));
                }
                active = false;
            }
            const int busy = __popcll(__ballot(active));
            if (busy == 0) break;
            if (moreWork && busy < tune.refillBelow) break;
        }
    }
    if (PROF) {
        for (int off = 32; off > 0; off >>= 1) { pfPush += __shfl_down(pfPush, off); pfSpill += __shfl_down(pfSpill, off); pfLeaf += __shfl_down(pfLeaf, off); pfLeafOk += __shfl_down(pfLeafOk, off); }
        if (lane == 0) {
            pf[0] = clock64() - pfStart; pf[11] = pfLeaf; pf[12] = pfLeafOk;
            for (int k = 0; k < 13; ++k) atomicAdd(&g_traceProf[(ANY_HIT ? 16 : 0) + k], pf[k]);
            atomicAdd(&g_traceProf[(ANY_HIT ? 16 : 0) + 13], 1ull);
            atomicAdd(&g_traceProf[(ANY_HIT ? 16 : 0) + 14], (unsigned long long)pfPush); atomicAdd(&g_traceProf[(ANY_HIT ? 16 : 0) + 15], (unsigned long long)pfSpill);
        }
    }
}

// ---------------------------------------------------------------------------
// k_generate: Sampler::GetCameraSample (core/sampler.cpp:46-52) + GenerateRayDifferential.
// path id = sampleInChunk * nPix + pixelIndex, so consecutive lanes are consecutive pixels
// of a 16x16 tile (coherent primary rays).  Bounce 0 uses the path id as stream index.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void footprint(const FilmGeom &fg, int px, int py, float fx, float fy, int *x0, int *x1, int *y0, int *y1) {
    // tile pixel bounds of the tile that samples (px,py): Film::GetFilmTile, core/film.cpp:96-107
    const int tx0 = fg.sx0 + ((px - fg.sx0) / 16) * 16, ty0 = fg.sy0 + ((py - fg.sy0) / 16) * 16;
    const int tx1 = sel_min(tx0 + 16, fg.sx1), ty1 = sel_min(ty0 + 16, fg.sy1);
    const int bx0 = sel_max((int)ceilf((float)tx0 - 0.5f - fg.rx), fg.cx0), by0 = sel_max((int)ceilf((float)ty0 - 0.5f - fg.ry), fg.cy0);
    const int bx1 = sel_min((int)floorf((float)tx1 - 0.5f + fg.rx) + 1, fg.cx1), by1 = sel_min((int)floorf((float)ty1 - 0.5f + fg.ry) + 1, fg.cy1);
    const float dxf = fx - 0.5f, dyf = fy - 0.5f;
    *x0 = sel_max((int)ceilf(dxf - fg.rx), bx0); *y0 = sel_max((int)ceilf(dyf - fg.ry), by0);
    *x1 = sel_min((int)floorf(dxf + fg.rx) + 1, bx1); *y1 = sel_min((int)floorf(dyf + fg.ry) + 1, by1);
}
__global__ __launch_bounds__(256) void k_generate(DevScene sc, RenderParams rp, PathStream out, uint32_t s0, uint32_t nSlots, IrregularSink irr) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    bool irregular = false;
    IrregularSample rec;
    if (slot < nSlots) {
        const uint32_t pix = slot % rp.nPix, sIdx = slot / rp.nPix;
        const uint32_t pxy = rp.pixelXY[pix];
        const int px = (int)(pxy & 0xffffu), py = (int)(pxy >> 16);
        const uint64_t index = (uint64_t)rp.pixelOffset[pix] + (uint64_t)(s0 + sIdx) * (uint64_t)rp.hal.sampleStride;
        const float u0 = halton_dim(sc, rp.hal, index, 0), u1 = halton_dim(sc, rp.hal, index, 1);
        const float fx = (float)px + u0, fy = (float)py + u1;
        float lu = 0.f, lv = 0.f;
        if (rp.cam.lensRadius > 0) { lu = halton_dim(sc, rp.hal, index, 3); lv = halton_dim(sc, rp.hal, index, 4); }
        DRay ray;
        camera_ray(rp.cam, fx, fy, lu, lv, &ray);
        out.ray.a[slot] = make_float4(ray.o.x, ray.o.y, ray.o.z, ray.tMax);
        out.ray.b[slot] = make_float4(ray.d.x, ray.d.y, ray.d.z, __uint_as_float(5u));   // sampler dimension 5 (after pFilm, time, pLens), bounce 0
        // beta = (1,1,1 | path id = slot) and L = (0,0,0 | continues) of a fresh path are not stored: k_bin and k_shade know them at bounce 0
        if (irr.count) {      // the film's irregular samples (see k_find_irregular, which this replaces in a render)
            int x0, x1, y0, y1;
            footprint(irr.fg, px, py, fx, fy, &x0, &x1, &y0, &y1);
            irregular = !(x0 == px && x1 == px + 1 && y0 == py && y1 == py + 1);
            rec.pix = pix; rec.sample = s0 + sIdx; rec.x0 = (int16_t)x0; rec.x1 = (int16_t)x1; rec.y0 = (int16_t)y0; rec.y1 = (int16_t)y1;
        }
    }
    if (irr.count) {      // (kernel-uniform)
        const uint32_t pos = wave_append(irr.count, irregular);
        if (irregular && pos < irr.capacity) irr.out[pos] = rec;
    }
}

// ---------------------------------------------------------------------------
// k_bin: material sorting.  After the closest-hit pass every active path is appended
// (ballot + mbcnt prefix, one atomic per wave and bin) to the queue of the shading
// variant it needs, so that a wavefront of k_shade runs one material's code:
//   bin 0  triangle with a matte material       (Lambertian lobe only)
//   bin 1  triangle with a plastic material     (Lambertian + Trowbridge-Reitz lobes)
//   bin 2  anything else that still needs work  (sphere hits: emitter / quadric fill)
// Paths that escaped the scene, or reached maxDepth (integrators/path.cpp:110), need no
// shading: they end here and hand their radiance to Lfinal[path id]; emitted radiance is
// only added at bounce 0 (path.cpp:97).
// ---------------------------------------------------------------------------
// Every workgroup bins HPRT_BIN_ITEMS x 1024 paths and appends to each bin with ONE atomic: single-address atomics retire
// at ~90 per microsecond on this chip, and with one workgroup per 1024 paths the three atomics of 122 k workgroups were most
// of this kernel's time (it moved 2.9 TB/s where plain stream kernels reach 5-6).  The counters of the bins sit on separate
// 256-byte lines (BIN_STRIDE).
#define HPRT_BIN_ITEMS 4
__global__ __launch_bounds__(1024) void k_bin(DevScene sc, PathStream in, HitStream hit, const uint32_t *queue, const uint32_t *countPtr,
                                              uint32_t countImm, int32_t maxDepth, int32_t bounces, BinSet bins, float4 *Lfinal) {
    __shared__ uint32_t waveCount[N_BINS][HPRT_BIN_ITEMS * 16];      // [bin][item round * 16 + wave]
    __shared__ uint32_t binBase[N_BINS];
    const uint32_t n = countPtr ? *countPtr : countImm;
    const uint32_t first = blockIdx.x * (HPRT_BIN_ITEMS * 1024u);
    if (first >= n) return;      // whole block beyond the queue (grids are sized for the batch)
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    int bin[HPRT_BIN_ITEMS];
    uint32_t slot[HPRT_BIN_ITEMS];
    float4 hitA[HPRT_BIN_ITEMS];
    // round k of the block covers the 1024 consecutive queue entries first + k * 1024 ...: all loads of a thread in flight together
#pragma unroll
    for (int k = 0; k < HPRT_BIN_ITEMS; ++k) {
        const uint32_t i = first + k * 1024u + threadIdx.x;
        slot[k] = i < n ? (queue ? queue[i] : i) : 0u;
    }
#pragma unroll
    for (int k = 0; k < HPRT_BIN_ITEMS; ++k) {
        const uint32_t i = first + k * 1024u + threadIdx.x;
        if (i < n) hitA[k] = hit.a[slot[k]];
    }
    unsigned long long mask[HPRT_BIN_ITEMS][N_BINS];
#pragma unroll
    for (int k = 0; k < HPRT_BIN_ITEMS; ++k) {
        const uint32_t i = first + k * 1024u + threadIdx.x;
        bin[k] = -1;
        if (i < n) {
            // the hit's primitive word carries the bin (dev_scene.h); every path of this pass is at the same bounce
            const int32_t word = __float_as_int(hitA[k].y);
            if (word >= 0) {
                // triangles reached directly go to the material-specialised variants; quadrics and hits inside
                // object instances (surface interaction transformed back to world space) to the generic one
                const int code = (int)(((uint32_t)word >> HIT_BIN_SHIFT) & 7u);
                const bool generic = code == (int)BIN_GENERIC || code == (int)BIN_TEXTURED;
                // at the depth limit only an emitter matters, hit by a camera ray or through a specular bounce (path.cpp:97-110)
                if (bounces >= maxDepth) { if (generic && (bounces == 0 || (__float_as_uint(in.ray.b[slot[k]].w) >> 31))) bin[k] = code; }
                else bin[k] = code;
            }
            // an escaped camera or specular segment picks up the infinite lights' radiance (integrators/path.cpp:97-106): the generic variant adds it
            if (word < 0 && sc.nEnvLights != 0u && (bounces == 0 || (__float_as_uint(in.ray.b[slot[k]].w) >> 31))) bin[k] = 2;
            if (bin[k] < 0) {      // the path ends here
                if (bounces == 0) Lfinal[slot[k]] = make_float4(0.f, 0.f, 0.f, 1.f);      // fresh path: path id = slot, L = 0 (k_generate)
                else {
                    const float4 b4 = in.beta[slot[k]];
                    float4 L4 = in.L[slot[k]];
                    // a specular segment that ends on a non-emitter at the depth limit still adds beta * isect.Le(-ray.d) = beta * 0 (path.cpp:97-100
                    // precedes the depth test, :110): +-0 unless the throughput is infinite or NaN — then NaN, and the reference zeroes the sample
                    if (word >= 0 && (__float_as_uint(in.ray.b[slot[k]].w) >> 31)) { L4.x += b4.x * 0.f; L4.y += b4.y * 0.f; L4.z += b4.z * 0.f; }
                    Lfinal[__float_as_uint(b4.w)] = L4;
                }
            }
        }
#pragma unroll
        for (int b = 0; b < (int)N_BINS; ++b) {
            mask[k][b] = __ballot(bin[k] == b);
            if (lane == 0) waveCount[b][k * 16 + wave] = (uint32_t)__popcll(mask[k][b]);
        }
    }
    __syncthreads();
    // exclusive scan of the 64 (round, wave) counts of each bin by one wave per bin, then one atomic per bin
    if (wave < N_BINS) {
        const uint32_t c = waveCount[wave][lane];
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(incl, off); if ((int)lane >= off) incl += t; }
        waveCount[wave][lane] = incl - c;
        if (lane == 63) binBase[wave] = incl ? atomicAdd(bins.count + wave * BIN_STRIDE, incl) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < HPRT_BIN_ITEMS; ++k) {
        if (bin[k] < 0) continue;
        const int b = bin[k];
        const unsigned long long m = b == 0 ? mask[k][0] : b == 1 ? mask[k][1] : b == 2 ? mask[k][2] : b == 3 ? mask[k][3] : mask[k][4];
        const uint32_t pos = binBase[b] + waveCount[b][k * 16 + wave] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        bins.q[b][pos] = slot[k];
    }
}

// ---------------------------------------------------------------------------
// k_shade<MODE>: one lane per path of bin MODE.  Consumes the closest hit of the path ray
// (input streams, gathered through the bin's ascending index list) and writes everything it
// produces at its own output index: bin 0 occupies [0, n0), bin 1 [n0, n0+n1), bin 2 follows.
// ---------------------------------------------------------------------------
// Specialised variants (MODE 0/1) run 512-thread workgroups (BS) so that queue appends cost one
// atomic per queue and workgroup; the rare generic variant keeps 256 threads (it needs more
// registers than a larger workgroup can have).
// TEX: scenes with image textures get their own instance of the generic variant (the lookups cost registers and a
// call stack that every other scene would pay for in occupancy).
// Waves per SIMD the register allocation of each variant is held to (second __launch_bounds__ argument): matte 111 and plastic
// 120 registers -> four, generic 164 -> three, which is what the allocator picks on its own; the knobs exist for A/B builds
// (tools/build_variant.sh).
#ifndef HPRT_SHADE_WAVES_MATTE
#define HPRT_SHADE_WAVES_MATTE 4
#endif
#ifndef HPRT_SHADE_WAVES_PLASTIC
#define HPRT_SHADE_WAVES_PLASTIC 4
#endif
#ifndef HPRT_SHADE_WAVES_GENERIC
#define HPRT_SHADE_WAVES_GENERIC 3
#endif
#ifndef HPRT_SHADE_WAVES_SUBSTRATE
#define HPRT_SHADE_WAVES_SUBSTRATE 4
#endif
// INSTS: the scene has object instances — the specialised variants then carry the instance transform as well (a hit inside an instance keeps
// its material's bin); without instances that code is compiled out of them.
template <int MODE, int BS, bool TEX = false, bool INSTS = false>
__global__ __launch_bounds__(BS, MODE == 0 ? HPRT_SHADE_WAVES_MATTE : MODE == 1 ? HPRT_SHADE_WAVES_PLASTIC : MODE == 3 ? HPRT_SHADE_WAVES_SUBSTRATE : HPRT_SHADE_WAVES_GENERIC) void k_shade(DevScene sc, RenderParams rp, PathStream in, HitStream hit, uint32_t s0,
                                               PathStream out, VertexStreams vs, QueueSet q, BinSet bins, float4 *Lfinal, uint32_t firstBounce, uint32_t retryPass) {
    __shared__ HaltonLds hl;
    __shared__ BlockAppendLds al;
    // MODE: the code variant — 0 matte, 1 plastic, 3 substrate (FresnelBlend only, materials/substrate.cpp:44-65), 2 generic; BIN: the bin it shades
    constexpr int BIN = MODE == 2 ? (TEX ? (int)BIN_TEXTURED : (int)BIN_GENERIC) : MODE == 3 ? (int)BIN_SUBSTRATE : MODE;
    constexpr int RETRY = BIN == (int)BIN_TEXTURED ? 1 : 0;      // voxel misses of the other bins are shaded again by the generic variant, the textured bin's by its own
    const uint32_t n = retryPass ? bins.count[(6 + RETRY) * BIN_STRIDE] : bins.count[BIN * BIN_STRIDE];
    if (blockIdx.x * blockDim.x >= n) return;      // whole block beyond the bin (grids are sized for the upper bound)
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
#ifdef HPRT_SHADE_PROF
    unsigned long long spT[8] = {0, 0, 0, 0, 0, 0, 0, 0}, spLast = clock64();
    bool spFull = false;
#endif
    bool wantNext = false, wantShadow = false, wantMis = false, wantResolve = false, defer = false, voxelMiss = false;
    uint32_t slot = 0;          // index in the input streams
    uint32_t j = 0;             // index in the output streams
    // The vertex's queue entry and path words are requested BEFORE the Halton tables are staged: the staging (20 KB per workgroup and a
    // barrier, in front of everything) and the first two round trips of the vertex's own dependent chain then overlap.
    float4 rayA = make_float4(0.f, 0.f, 0.f, 0.f), rayB = rayA, hitA = rayA, beta4 = rayA, L4 = rayA;
    if (i < n) {
        if (retryPass) { const uint2 e = bins.retry[RETRY][i]; slot = e.x; j = e.y; }
        else {
        slot = bins.q[BIN][i];
        // output index: bins in the order matte, plastic, substrate, generic (as binned), textured
        j = (BIN == (int)BIN_MATTE ? 0u : BIN == (int)BIN_PLASTIC ? bins.count[0] : BIN == (int)BIN_SUBSTRATE ? bins.count[0] + bins.count[BIN_STRIDE]
             : BIN == (int)BIN_GENERIC ? bins.count[0] + bins.count[BIN_STRIDE] + bins.count[4 * BIN_STRIDE]
             : bins.count[0] + bins.count[BIN_STRIDE] + bins.count[4 * BIN_STRIDE] + bins.count[5 * BIN_STRIDE]) + i;
        if (BIN == (int)BIN_GENERIC && i >= bins.count[5 * BIN_STRIDE]) j = bins.aux[i];       // deferred by a specialised variant: keeps that variant's index
        }
        rayA = in.ray.a[slot]; rayB = in.ray.b[slot]; hitA = hit.a[slot];
        // a fresh path's throughput and radiance are constants (k_generate does not store them)
        beta4 = firstBounce ? make_float4(1.f, 1.f, 1.f, __uint_as_float(slot)) : in.beta[slot];
        L4 = firstBounce ? make_float4(0.f, 0.f, 0.f, 1.f) : in.L[slot];
    }
    halton_lds_load(sc, &hl);
    if (i < n) {
        SP_MARK(6);      // queue entry + path streams
        const uint32_t st = __float_as_uint(rayB.w);
        int dim = (int)(st & 0xffffu);
        const int bounces = (int)((st >> 16) & 0x7fffu);
        const bool specularBounce = (st >> 31) != 0u;      // the segment left a specular lobe (integrators/path.cpp:160)
        const uint32_t pathId = __float_as_uint(beta4.w);
        const uint32_t pix = pathId % rp.nPix, sIdx = pathId / rp.nPix;
        const uint64_t index = (uint64_t)rp.pixelOffset[pix] + (uint64_t)(s0 + sIdx) * (uint64_t)rp.hal.sampleStride;
        const int32_t prim = hit_prim(__float_as_int(hitA.y));
        const vec3 rayO(rayA.x, rayA.y, rayA.z);
        const vec3 rayD(rayB.x, rayB.y, rayB.z);
        rgb beta(beta4.x, beta4.y, beta4.z);
        rgb L(L4.x, L4.y, L4.z);
        float etaScale = L4.w;      // (a stored path continues, so its w word is its etaScale; fresh paths: 1)
        const bool found = prim >= 0;
        DevSI si;
        DevTexGeom tg;
        constexpr bool texScene = MODE == 2 && TEX;      // image textures: the generic variant only
        if (found) {
            const float4 v0 = sc.tris[3 * prim];
            const float2 hitB = hit.b[slot];
            // A hit inside an object instance was found by the instance-space ray: the surface interaction is
            // filled there and transformed back (TransformedPrimitive::Intersect, core/primitive.cpp:77-93)
            const int inst = (MODE == 2 || INSTS) ? __float_as_int(hitB.y) : -1;
            DRay r0; r0.o = rayO; r0.d = rayD; r0.tMax = rayA.w;
            if ((MODE == 2 || INSTS) && inst >= 0) {      // Transform::operator()(const Ray &), core/transform.h:251-264
                const DevInstance &in = sc.instances[inst];
                vec3 oErr;
                r0.o = xf_point_err(in.w2i, rayO, &oErr);
                r0.d = xf_vector(in.w2i, rayD);
                const float lengthSquared = r0.d.x * r0.d.x + r0.d.y * r0.d.y + r0.d.z * r0.d.z;
                if (lengthSquared > 0) {
                    const float dt = dot(vabs(r0.d), oErr) / lengthSquared;
                    r0.o = r0.o + r0.d * dt;
                    r0.tMax -= dt;
                }
            }
            if (MODE != 2 || (__float_as_uint(v0.w) & TAG_KIND_MASK) == 0u)
                fill_triangle(sc, (uint32_t)prim, hitA.z, hitA.w, hitB.x, r0.d, &si, texScene ? &tg : nullptr);
            else {
                float tt;
                fill_sphere(sc, (int)__float_as_uint(sc.tris[3 * prim + 1].w), r0, &si, &tt, texScene ? &tg : nullptr);
            }
            if ((MODE == 2 || INSTS) && inst >= 0 && !sc.instances[inst].identity) {
                // Transform::operator()(const SurfaceInteraction &), core/transform.cpp:262-297
                const DevInstance &in = sc.instances[inst];
                vec3 pErr;
                si.p = xf_point_err_in(in.i2w, si.p, si.pErr, &pErr); si.pErr = pErr;
                si.n = normalize(xf_normal(in.w2i, si.n));
                si.wo = normalize(xf_vector(in.i2w, si.wo));
                si.ns = normalize(xf_normal(in.w2i, si.ns));
                si.sdpdu = xf_vector(in.i2w, si.sdpdu);
                si.ns = face_forward(si.ns, si.n);
                if (texScene) { tg.dpdu = xf_vector(in.i2w, tg.dpdu); tg.dpdv = xf_vector(in.i2w, tg.dpdv); }
            }
            // emitted radiance at the first vertex (path.cpp:97-107; no specular lobes exist here).
            // Emissive triangles carry TAG_GENERIC (like quadrics), so only the generic variant looks.
            if (MODE == 2 && (bounces == 0 || specularBounce)) {
                const int al = prim_area_light(sc, prim);
                if (al >= 0) {
                    rgb Le = area_L(sc.lights[al], si.n, -rayD);
                    rgb add = beta * Le;
                    L = rgb(L.r + add.r, L.g + add.g, L.b + add.b);
                } else if (specularBounce) {
                    // L += beta * Spectrum(0) (path.cpp:97-100 with isect.Le == 0): +-0 for a finite throughput — L unchanged — but NaN where a
                    // specular bounce left beta infinite or NaN, and the reference then zeroes the whole sample (core/integrator.cpp:300-321)
                    L = rgb(L.r + beta.r * 0.f, L.g + beta.g * 0.f, L.b + beta.b * 0.f);
                }
            } else if (MODE != 2 && specularBounce) L = rgb(L.r + beta.r * 0.f, L.g + beta.g * 0.f, L.b + beta.b * 0.f);      // (the same term on a non-emitter reached directly; beta == 1 at bounce 0)
        }
        SP_MARK(0);      // loads + surface interaction
        if (MODE == 2 && !found && (bounces == 0 || specularBounce))
            for (uint32_t e = 0; e < sc.nEnvLights; ++e) { const rgb add = beta * env_Le(sc, sc.envLights[e], rayD); L = rgb(L.r + add.r, L.g + add.g, L.b + add.b); }
        if (found && bounces < rp.maxDepth) {
#ifdef HPRT_SHADE_PROF
            spFull = true;
#endif
            DevBsdf bsdf;
            bool useKd = false, useKs = false, useOp = false;
            rgb kdTex, ksTex, opTex;
            if (texScene) {
                const DevMaterial m = sc.materials[sc.shapes[si.shape].material];
                const int opTexId = m.type == 6 ? m.opTex : -1;
                if (m.KdTex >= 0 || m.KsTex >= 0 || opTexId >= 0) {
                    // SurfaceInteraction::ComputeDifferentials (core/interaction.cpp:103-149): only the camera ray carries
                    // differentials (RayDifferential(const Ray &) clears them for every spawned ray, core/geometry.h:1213-1216)
                    DevUvDiff uvd; uvd.dudx = uvd.dvdx = uvd.dudy = uvd.dvdy = 0.f;
                    if (bounces == 0) {
                        const float u0 = halton_dim(sc, rp.hal, index, 0, &hl), u1 = halton_dim(sc, rp.hal, index, 1, &hl);
                        const uint32_t pxy = rp.pixelXY[pix];
                        const float fx = (float)(int)(pxy & 0xffffu) + u0, fy = (float)(int)(pxy >> 16) + u1;
                        float lu = 0.f, lv = 0.f;
                        if (rp.cam.lensRadius > 0) { lu = halton_dim(sc, rp.hal, index, 3, &hl); lv = halton_dim(sc, rp.hal, index, 4, &hl); }
                        DevRayDiff rd;
                        camera_ray_diff(rp.cam, fx, fy, lu, lv, &rd);
                        const float s = rp.invSqrtSpp;      // RayDifferential::ScaleDifferentials, core/geometry.h:1222-1227
                        rd.rxO = rayO + (rd.rxO - rayO) * s; rd.ryO = rayO + (rd.ryO - rayO) * s;
                        rd.rxD = rayD + (rd.rxD - rayD) * s; rd.ryD = rayD + (rd.ryD - rayD) * s;
                        compute_differentials(si, tg, rd, &uvd);
                    }
                    // one inlined copy of the lookup for the three slots (three call sites of an out-of-line lookup cost a register spill each)
#pragma unroll 1
                    for (int k = 0; k < 3; ++k) {
                        const int id = k == 0 ? m.KdTex : k == 1 ? m.KsTex : opTexId;
                        if (id < 0) continue;
                        const rgb r = eval_image_texture(sc, id, tg, uvd);
                        if (k == 0) { kdTex = r; useKd = true; } else if (k == 1) { ksTex = r; useKs = true; } else { opTex = r; useOp = true; }
                    }
                }
            }
            bsdf_init(sc, si, &bsdf, useKd ? &kdTex : nullptr, useKs ? &ksTex : nullptr, useOp ? &opTex : nullptr);
            if (MODE == 0) { bsdf.hasS = false; bsdf.Rs = rgb(0.f); bsdf.alpha = 0.f; }   // matte: no microfacet lobe (matte.cpp:45-62)
            if (MODE != 2) { bsdf.hasR = false; bsdf.hasT = false; bsdf.oren = false; bsdf.kind = 0; }      // mirror, metal, glass, uber and OrenNayar surfaces are shaded by the generic variant
            if (MODE == 3) { bsdf.kind = 2; bsdf.hasD = false; }      // substrate: the one FresnelBlend lobe (bsdf_init set hasS unless both reflectances are black)
            SP_MARK(1);      // textures + bsdf_init
            // ---- direct lighting (UniformSampleOneLight + EstimateDirect) ----
            if (bsdf_num(bsdf) > 0 && sc.nLights > 0) {
                // The reference draws five sample values here (light pick, uLight, uScattering: core/integrator.cpp:96-104); a value
                // that cannot influence anything is not evaluated, its dimension is consumed all the same: with a single light
                // every pick value selects light 0 with pdf 1, and point / distant lights ignore uLight (lights/point.cpp:44-53,
                // lights/distant.cpp:49-59) and never reach the BSDF-sampling branch (core/integrator.cpp:168).
                float pickPdf;
                const int lightNum = light_pick(sc, si.p, sc.nLights > 1u ? halton_dim(sc, rp.hal, index, dim, &hl) : 0.f, &pickPdf, &voxelMiss);
                dim += 1;
                if (voxelMiss) defer = true;      // nothing has been written for this vertex: it is shaded again once its voxel is there
                if (pickPdf != 0) {
                    const DevLight light = sc.lights[lightNum];
                    const bool isDelta = light.type < 2;
                    float ul0 = 0.f, ul1 = 0.f, us0 = 0.f, us1 = 0.f;
                    if (!isDelta) {
                        ul0 = halton_dim(sc, rp.hal, index, dim, &hl); ul1 = halton_dim(sc, rp.hal, index, dim + 1, &hl);
                        us0 = halton_dim(sc, rp.hal, index, dim + 2, &hl); us1 = halton_dim(sc, rp.hal, index, dim + 3, &hl);
                    }
                    dim += 4;
                    vec3 wi;
                    float lightPdf = 0, scatteringPdf = 0;
                    DevIt it; it.p = si.p; it.pErr = si.pErr; it.n = si.n;
                    DevIt pl;
                    // A shading point inside the emitter sphere needs the full quadric code
                    // (Sphere::Sample(u) + Shape::Pdf); nothing has been written yet, so the
                    // material-specialised variants hand such a vertex to the generic variant.
                    // (likewise a vertex lit by a triangle emitter: Triangle::Sample / Shape::Pdf live in the generic variant only)
                    if (MODE != 2 && !isDelta && (light.type >= 3 || sphere_ref_inside(sc.spheres[light.sphere], it))) defer = true;      // (type 4, an infinite light: generic variant too)
                    SP_MARK(2);      // light pick + the four sample values
                    if (!defer) {
                    rgb Li = light_sample<MODE == 2>(sc, light, it, ul0, ul1, &wi, &lightPdf, &pl);
                    rgb pendLight(0.f), pendMis(0.f);
                    if (lightPdf > 0 && !is_black(Li)) {
                        rgb f = bsdf_f(bsdf, si.wo, wi) * absdot(wi, si.ns);
                        scatteringPdf = bsdf_pdf(bsdf, si.wo, wi);
                        if (!is_black(f)) {
                            // shadow ray: Interaction::SpawnRayTo(it) (core/interaction.h:73-78)
                            vec3 origin = offset_ray_origin(it.p, it.pErr, it.n, pl.p - it.p);
                            vec3 target = offset_ray_origin(pl.p, pl.pErr, pl.n, origin - pl.p);
                            vec3 d = target - origin;
                            vs.shadow.a[j] = make_float4(origin.x, origin.y, origin.z, 1 - HPRT_SHADOW_EPS);
                            vs.shadow.b[j] = make_float4(d.x, d.y, d.z, 0.f);
                            if (isDelta) pendLight = f * Li / lightPdf;
                            else { float w = power_heuristic(lightPdf, scatteringPdf); pendLight = f * Li * w / lightPdf; }
                            wantShadow = true;
                        }
                    }
                    SP_MARK(3);      // light sample + f + pdf + shadow ray
                    if (!isDelta) {
                        int sampledType = 0;
                        rgb f = bsdf_sample(bsdf, si.wo, &wi, us0, us1, &scatteringPdf, &sampledType);
                        f = f * absdot(wi, si.ns);
                        if (!is_black(f) && scatteringPdf > 0) {
                            // EstimateDirect traces this ray only to learn whether its closest hit is the emitter
                            // (core/integrator.cpp:176-190); Sphere::Pdf is non-zero for every direction, so most of these
                            // rays point nowhere near it.  The quadric pre-test (dev_intersect.h) proves for a ray that the
                            // emitter's Intersect returns false whatever tMax is; such a ray adds exactly nothing and is
                            // not traced.  (Counting renders trace it anyway unless told otherwise: the reference counts it.)
                            const vec3 o = offset_ray_origin(si.p, si.pErr, si.n, wi);
                            DRay mr; mr.o = o; mr.d = wi; mr.tMax = HPRT_INF;
                            // (triangle emitters: the traversal's own triangle test on the emitter with tMax = infinity)
                            float tb0, tb1, tb2, tt;
                            // (an infinite light is reached by every such ray that escapes: nothing to cull)
                            if (!rp.cullMis || (MODE == 2 && light.type == 4) ||
                                (MODE == 2 && light.type == 3 ? triangle_may_hit(sc, light.prim, mr, &tb0, &tb1, &tb2, &tt)
                                                              : sphere_may_hit(sc.spheres[light.sphere], mr))) {
                                const float lp = light_pdf<MODE == 2>(sc, light, it, wi);
                                if (lp != 0) {     // "if (lightPdf == 0) return Ld;" keeps the light-sampling term only
                                    const float w = power_heuristic(scatteringPdf, lp);
                                    vs.mis.a[j] = make_float4(o.x, o.y, o.z, HPRT_INF);
                                    vs.mis.b[j] = make_float4(wi.x, wi.y, wi.z, 0.f);
                                    // contribution if the ray reaches the light's emitting side: f * Li * Tr * weight / pdf
                                    rgb Lemit(light.I[0], light.I[1], light.I[2]);
                                    if (MODE == 2 && light.type == 4) Lemit = env_Le(sc, sc.envLights[light.shape], wi);      // light.Le(ray) of the escaped ray (core/integrator.cpp:201-202)
                                    pendMis = f * Lemit * rgb(1.f) * w / scatteringPdf;
                                    wantMis = true;
                                }
                            }
                        }
                    }
                    if (wantShadow || wantMis) {
                        vs.pendLight[j] = make_float4(pendLight.r, pendLight.g, pendLight.b,
                                                      __uint_as_float((uint32_t)lightNum | (wantShadow ? 0x40000000u : 0u) | (wantMis ? 0x80000000u : 0u)));
                        // (k_resolve re-derives pickPdf from the light number — except with the spatial distribution, where it depends on the vertex)
                        if (wantMis || sc.spatial) vs.pendMis[j] = make_float4(pendMis.r, pendMis.g, pendMis.b, pickPdf);
                        vs.pendBeta[j] = make_float4(beta.r, beta.g, beta.b, beta4.w);
                        wantResolve = true;
                    }
                    // neither ray: Ld == 0, "L += beta * 0 / pdf" leaves L unchanged
                    }
                }
            }
            // "L += beta * UniformSampleOneLight(...)" (path.cpp:129-137) runs for every vertex with a non-specular lobe, also when the estimate
            // is Spectrum(0) (no light, pick pdf 0, black f, zero light pdf): beta * 0 is +-0 for a finite throughput — which is why such a vertex
            // queues no resolve work — but NaN for an infinite or NaN one (a degenerate microfacet alpha, a vanishing pdf), and the
            // reference's NaN guard then zeroes the whole sample.  Found by the random scenes (rough glass with alpha 0 along one axis).
            if (!defer && !wantResolve && bsdf_num(bsdf) > 0) L = rgb(L.r + beta.r * 0.f, L.g + beta.g * 0.f, L.b + beta.b * 0.f);
            SP_MARK(4);      // BSDF-sampled light term + pending stores
            // ---- sample the BSDF for the next path segment (path.cpp:141-164) ----
            if (!defer) {
            const float ub0 = halton_dim(sc, rp.hal, index, dim, &hl), ub1 = halton_dim(sc, rp.hal, index, dim + 1, &hl);
            dim += 2;
            vec3 wo = -rayD, wi;
            float pdf = 0; int flags = 0;
            rgb f = bsdf_sample(bsdf, wo, &wi, ub0, ub1, &pdf, &flags, true);
            if (!(is_black(f) || pdf == 0.f)) {
                beta = beta * (f * absdot(wi, si.ns) / pdf);
                vec3 o = offset_ray_origin(si.p, si.pErr, si.n, wi);
                bool alive = true;
                // Russian roulette (path.cpp:191-199); etaScale == 1 without transmission
                // (etaScale travels in the w word of the radiance stream: 1 until a path crosses a dielectric boundary, path.cpp:154-162)
                if (MODE == 2 && (flags & BX_SPECULAR) && (flags & BX_TRANSMISSION)) {
                    const float eta = bsdf.eta;
                    etaScale *= (dot(wo, si.n) > 0) ? (eta * eta) : 1 / (eta * eta);
                }
                rgb rrBeta = beta * etaScale;
                if (max_value(rrBeta) < rp.rrThreshold && bounces > 3) {
                    float qv = sel_max(.05f, 1 - max_value(rrBeta));
                    float u = halton_dim(sc, rp.hal, index, dim, &hl);
                    dim += 1;
                    if (u < qv) alive = false;
                    else beta = beta / (1 - qv);
                }
                // The reference also traces the segment that leaves the path's last vertex (bounces == maxDepth) and then
                // stops without using its hit: no emission is added after a non-specular bounce (path.cpp:97-110).  A plain
                // render does not trace that ray; a counting render does, the reference counts it.
                // (after a specular bounce the emitted radiance of that hit IS added, path.cpp:97: the segment is traced)
                if (rp.cullMis && bounces + 1 >= rp.maxDepth && !(flags & BX_SPECULAR)) alive = false;
                if (alive) {
                    out.ray.a[j] = make_float4(o.x, o.y, o.z, HPRT_INF);
                    out.ray.b[j] = make_float4(wi.x, wi.y, wi.z, __uint_as_float((uint32_t)dim | ((uint32_t)(bounces + 1) << 16) | ((flags & BX_SPECULAR) ? 0x80000000u : 0u)));
                    out.beta[j] = make_float4(beta.r, beta.g, beta.b, beta4.w);
                    wantNext = true;
                }
            }
            }
        }
        // Radiance so far: k_resolve adds this vertex's direct lighting to out.L[j] and, if the path
        // stops here (w == 0), passes the sum on to Lfinal; without a pending term this lane does it.
        if (!defer) {
            if (wantNext || wantResolve) out.L[j] = make_float4(L.r, L.g, L.b, wantNext ? etaScale : 0.f);      // w: 0 = the path ends at this vertex, else its etaScale
            else Lfinal[pathId] = make_float4(L.r, L.g, L.b, 0.f);
        }
    }
    SP_MARK(5);      // next segment: sample values, bsdf_sample, roulette, stream stores
#ifdef HPRT_SHADE_PROF
    {
        const unsigned long long m = __ballot(spFull);
        if (m != 0ull && __lane_id() == (uint32_t)(__ffsll((long long)m) - 1)) {
            for (int k = 0; k < 7; ++k) atomicAdd(&g_shadeProf[MODE * 8 + k], spT[k]);
            atomicAdd(&g_shadeProf[MODE * 8 + 7], 1ull);
        }
    }
#endif
    // queue appends in block-uniform control flow
    if (MODE != 2) {   // (almost) never
        const bool toGeneric = defer && !voxelMiss;
        const uint32_t p2 = wave_append(bins.count + 2 * BIN_STRIDE, toGeneric);
        if (toGeneric) { bins.q[2][p2] = slot; bins.aux[p2] = j; }
    }
    if (sc.voxSlot) {   // (kernel-uniform) on-demand voxel tables: the vertices that missed, for the pass after the fill
        const uint32_t pr = wave_append(bins.count + (6 + RETRY) * BIN_STRIDE, voxelMiss && !retryPass);
        if (voxelMiss && !retryPass) bins.retry[RETRY][pr] = make_uint2(slot, j);
    }
    uint32_t *const ctr[4] = {q.nextCount, q.shadowCount, q.misCount, q.resolveCount};
    const bool pred[4] = {wantNext, wantShadow, wantMis, wantResolve};
    uint32_t pos[4];
    block_append<4>(&al, ctr, pred, pos);
    if (wantNext) q.next[pos[0]] = j;
    if (wantShadow) q.shadow[pos[1]] = j;
    if (wantMis) q.mis[pos[2]] = j;
    if (wantResolve) q.resolve[pos[3]] = j;
}

// ---------------------------------------------------------------------------
// k_resolve: Ld = [unoccluded ? light term] + [MIS ray reached the light ? bsdf term];
// L += beta * (Ld / lightPickPdf)   (core/integrator.cpp:106, integrators/path.cpp:132-137)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resolve(DevScene sc, VertexStreams vs, float4 *Lio, float4 *Lfinal, const uint32_t *queue,
                                                 const uint32_t *countPtr) {
    const uint32_t n = *countPtr;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = queue[i];
    const float4 pl = vs.pendLight[j], pb = vs.pendBeta[j], L4 = Lio[j];
    const uint32_t info = __float_as_uint(pl.w);
    const int lightNum = (int)(info & 0x3fffffffu);
    // the pending BSDF-sampled term exists for few vertices (most such rays are never queued): read it only then
    float4 pm = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((info & 0x80000000u) || sc.spatial) pm = vs.pendMis[j];
    // lightPdf of UniformSampleOneLight, as light_pick computed it for this light (core/integrator.cpp:94-99); the spatial
    // distribution's depends on the vertex's voxel and travels with the pending term
    const float pickPdf = sc.spatial ? pm.w : ((sc.lightFuncInt > 0) ? sc.lightFunc[lightNum] / (sc.lightFuncInt * (int)sc.nLights) : 0);
    rgb Ld(0.f);
    if ((info & 0x40000000u) && !vs.occluded[j]) Ld = Ld + rgb(pl.x, pl.y, pl.z);
    if (info & 0x80000000u) {
        const float4 mh = vs.misHit.a[j];
        const int32_t prim = hit_prim(__float_as_int(mh.y));
        if (prim < 0) {      // the ray escaped: an infinite light's radiance along it was priced in by k_shade, any other light has none (core/light.cpp:66)
            if (sc.lights[lightNum].type == 4) Ld = Ld + rgb(pm.x, pm.y, pm.z);
        } else if (prim_area_light(sc, prim) == lightNum) {
            // lightIsect.Le(-wi): the emitter's normal at the hit
            const float4 ma = vs.mis.a[j], mb = vs.mis.b[j];
            DRay r; r.o = vec3(ma.x, ma.y, ma.z); r.d = vec3(mb.x, mb.y, mb.z); r.tMax = HPRT_INF;
            const DevLight light = sc.lights[lightNum];
            DevSI li;
            bool filled;
            if (light.type == 3) { fill_triangle(sc, (uint32_t)prim, mh.z, mh.w, vs.misHit.b[j].x, r.d, &li); filled = true; }
            else { float tt; filled = fill_sphere(sc, (int)__float_as_uint(sc.tris[3 * prim + 1].w), r, &li, &tt); }
            if (filled && (light.twoSided || dot(li.n, -r.d) > 0)) Ld = Ld + rgb(pm.x, pm.y, pm.z);
        }
    }
    rgb add = rgb(pb.x, pb.y, pb.z) * (Ld / pickPdf);
    const float4 Lnew = make_float4(L4.x + add.r, L4.y + add.g, L4.z + add.b, L4.w);
    if (L4.w != 0.f) Lio[j] = Lnew;                                   // the path goes on: radiance travels with it
    else Lfinal[__float_as_uint(pb.w)] = Lnew;                        // last vertex of the path
}

// ---------------------------------------------------------------------------
// k_store_radiance: radiance guards (core/integrator.cpp:300-321) and transfer of the
// finished batch into the per-sample radiance store Lall[channel][sample][pixel].
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_store_radiance(const float4 *Lfinal, float *LallR, float *LallG, float *LallB, uint32_t nPix,
                                                         uint32_t s0, uint32_t nSlots) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= nSlots) return;
    const float4 L4 = Lfinal[slot];
    rgb L(L4.x, L4.y, L4.z);
    const float y = luminance(L);
    if (is_nan(L.r) || is_nan(L.g) || is_nan(L.b)) L = rgb(0.f);
    else if ((double)y < -1e-5) L = rgb(0.f);
    else if (is_inf(y)) L = rgb(0.f);
    const size_t o = (size_t)s0 * nPix + slot;   // slot = sIdx*nPix + pix
    LallR[o] = L.r; LallG[o] = L.g; LallB[o] = L.b;
}

// ---------------------------------------------------------------------------
// Film.  k_find_irregular lists the camera samples whose box-filter footprint is not
// exactly their own pixel (pFilm fraction 0, or pixel + u rounding up to the next
// integer: FilmTile::AddSample, core/film.h:136-143).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_find_irregular(DevScene sc, RenderParams rp, FilmGeom fg, uint32_t spp, uint32_t *count,
                                                        uint32_t capacity, IrregularSample *out) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)rp.nPix * spp;
    bool irregular = false;
    IrregularSample rec;
    if (g < total) {
        const uint32_t pix = (uint32_t)(g % rp.nPix), s = (uint32_t)(g / rp.nPix);
        const uint32_t pxy = rp.pixelXY[pix];
        const int px = (int)(pxy & 0xffffu), py = (int)(pxy >> 16);
        const uint64_t index = (uint64_t)rp.pixelOffset[pix] + (uint64_t)s * (uint64_t)rp.hal.sampleStride;
        const float fx = (float)px + halton_dim(sc, rp.hal, index, 0), fy = (float)py + halton_dim(sc, rp.hal, index, 1);
        int x0, x1, y0, y1;
        footprint(fg, px, py, fx, fy, &x0, &x1, &y0, &y1);
        irregular = !(x0 == px && x1 == px + 1 && y0 == py && y1 == py + 1);
        rec.pix = pix; rec.sample = s; rec.x0 = (int16_t)x0; rec.x1 = (int16_t)x1; rec.y0 = (int16_t)y0; rec.y1 = (int16_t)y1;
    }
    const uint32_t pos = wave_append(count, irregular);
    if (irregular && pos < capacity) out[pos] = rec;
}

__device__ __forceinline__ rgb clamp_luminance(rgb L, float maxY) {   // core/film.h:133-134
    float y = luminance(L);
    if (y > maxY) { float s = maxY / luminance(L); L = L * s; }
    return L;
}
__device__ __forceinline__ void rgb_to_xyz(rgb c, float xyz[3]) {     // core/spectrum.h:62-66
    xyz[0] = 0.412453f * c.r + 0.357580f * c.g + 0.180423f * c.b;
    xyz[1] = 0.212671f * c.r + 0.715160f * c.g + 0.072169f * c.b;
    xyz[2] = 0.019334f * c.r + 0.119193f * c.g + 0.950227f * c.b;
}
// One thread per local pixel: contribSum = pre-extras, own samples, post-extras, in the
// order the reference's tile loop produces them; then MergeFilmTile's RGB->XYZ.
__global__ __launch_bounds__(256) void k_film_own(RenderParams rp, FilmGeom fg, const float *LallR, const float *LallG,
                                                  const float *LallB, uint32_t spp, FilmExtras ex, float *filmXYZW) {
    const uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= rp.nPix) return;
    rgb c(0.f); float w = 0.f;
    const uint32_t eBeg = ex.ownBegin ? ex.ownBegin[pix] : 0u, eEnd = ex.ownBegin ? ex.ownBegin[pix + 1] : 0u;
    uint32_t e = eBeg;
    for (; e < eEnd && ex.ownIsPre[e]; ++e) {
        const size_t o = (size_t)ex.ownSample[e] * rp.nPix + ex.ownSrcPix[e];
        c = c + clamp_luminance(rgb(LallR[o], LallG[o], LallB[o]), fg.maxSampleLuminance) * 1.0f * 1.0f;
        w += 1.f;
    }
    for (uint32_t s = 0; s < spp; ++s) {
        const size_t o = (size_t)s * rp.nPix + pix;
        c = c + clamp_luminance(rgb(LallR[o], LallG[o], LallB[o]), fg.maxSampleLuminance) * 1.0f * 1.0f;
        w += 1.f;
    }
    for (; e < eEnd; ++e) {
        const size_t o = (size_t)ex.ownSample[e] * rp.nPix + ex.ownSrcPix[e];
        c = c + clamp_luminance(rgb(LallR[o], LallG[o], LallB[o]), fg.maxSampleLuminance) * 1.0f * 1.0f;
        w += 1.f;
    }
    float xyz[3];
    rgb_to_xyz(c, xyz);
    const uint32_t pxy = rp.pixelXY[pix];
    const int px = (int)(pxy & 0xffffu), py = (int)(pxy >> 16);
    const size_t fo = 4 * ((size_t)(py - fg.cy0) * (size_t)(fg.cx1 - fg.cx0) + (size_t)(px - fg.cx0));
    // Film::pixels start at zero: 0 + v
    filmXYZW[fo] = 0.f + xyz[0]; filmXYZW[fo + 1] = 0.f + xyz[1]; filmXYZW[fo + 2] = 0.f + xyz[2]; filmXYZW[fo + 3] = 0.f + w;
}
// One thread per destination pixel that receives samples from OTHER tiles: each source
// tile's FilmTile pixel is folded on its own, converted to XYZ and merged.
__global__ __launch_bounds__(64) void k_film_foreign(RenderParams rp, FilmGeom fg, const float *LallR, const float *LallG,
                                                     const float *LallB, FilmExtras ex, float *filmXYZW) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= ex.nForeignDest) return;
    const size_t fo = 4 * (size_t)ex.foreignDestFilmIndex[d];
    for (uint32_t g = ex.foreignDestBegin[d]; g < ex.foreignDestBegin[d + 1]; ++g) {
        rgb c(0.f); float w = 0.f;
        for (uint32_t e = ex.foreignGroupBegin[g]; e < ex.foreignGroupBegin[g + 1]; ++e) {
            const size_t o = (size_t)ex.foreignSample[e] * rp.nPix + ex.foreignSrcPix[e];
            c = c + clamp_luminance(rgb(LallR[o], LallG[o], LallB[o]), fg.maxSampleLuminance) * 1.0f * 1.0f;
            w += 1.f;
        }
        float xyz[3];
        rgb_to_xyz(c, xyz);
        filmXYZW[fo] += xyz[0]; filmXYZW[fo + 1] += xyz[1]; filmXYZW[fo + 2] += xyz[2]; filmXYZW[fo + 3] += w;
    }
}

// HPRT_RENDER_EXPORT_FOREIGN (tile-sharded renders): the same per-source-tile folds as k_film_foreign, but each
// group's {xyz, weight} is written out as a record instead of being merged, so that the gather can merge the groups of
// ALL ranks into the destination pixel in ascending source-tile order — the order of the single-GPU film.
__global__ __launch_bounds__(64) void k_film_foreign_export(RenderParams rp, FilmGeom fg, const float *LallR, const float *LallG,
                                                            const float *LallB, FilmExtras ex, uint32_t nGroups, const uint32_t *groupDest,
                                                            const uint32_t *groupTile, FilmRecord *out) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nGroups) return;
    rgb c(0.f); float w = 0.f;
    for (uint32_t e = ex.foreignGroupBegin[g]; e < ex.foreignGroupBegin[g + 1]; ++e) {
        const size_t o = (size_t)ex.foreignSample[e] * rp.nPix + ex.foreignSrcPix[e];
        c = c + clamp_luminance(rgb(LallR[o], LallG[o], LallB[o]), fg.maxSampleLuminance) * 1.0f * 1.0f;
        w += 1.f;
    }
    FilmRecord r;
    r.dest = groupDest[g]; r.srcTile = groupTile[g];
    rgb_to_xyz(c, r.xyz);
    r.w = w;
    out[g] = r;
}
// Film::MergeFilmTile's `mergePixel.xyz[i] += xyz[i]; filterWeightSum += ...` (core/film.cpp:124-131) for exported
// records: one thread per destination pixel, its records in ascending source-tile order.
__global__ __launch_bounds__(64) void k_film_apply_records(const FilmRecord *rec, const uint32_t *destBegin, uint32_t nDest, float *filmXYZW) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= nDest) return;
    const size_t fo = 4 * (size_t)rec[destBegin[d]].dest;
    for (uint32_t k = destBegin[d]; k < destBegin[d + 1]; ++k) {
        filmXYZW[fo] += rec[k].xyz[0]; filmXYZW[fo + 1] += rec[k].xyz[1]; filmXYZW[fo + 2] += rec[k].xyz[2]; filmXYZW[fo + 3] += rec[k].w;
    }
}

// SpatialLightDistribution::ComputeDistribution (core/lightdistrib.cpp:231-298) for EVERY voxel, at scene creation.
// k_voxel_contrib: one thread per (voxel, light): the 128 Halton points of the voxel (ri: RadicalInverse(0..4, i), [5][128], from the
// host), each light sampled from each point with the point's (ri[3], ri[4]); lightContrib += Li.y() / pdf in sample order.
// voxList (on-demand mode): row k of the output belongs to voxel voxList[k]; null: row = voxel
__global__ __launch_bounds__(256) void k_voxel_contrib(DevScene sc, const float *ri, uint32_t nVox, float *contrib, const uint32_t *voxList) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = sc.nLights;
    if (g >= (size_t)nVox * n) return;
    const uint32_t row = (uint32_t)(g / n), j = (uint32_t)(g % n);
    const uint32_t v = voxList ? voxList[row] : row;
    int pi[3]; pi[2] = (int)(v % (uint32_t)sc.voxN[2]); pi[1] = (int)((v / (uint32_t)sc.voxN[2]) % (uint32_t)sc.voxN[1]); pi[0] = (int)(v / ((uint32_t)sc.voxN[2] * (uint32_t)sc.voxN[1]));
    float vMin[3], vMax[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float t0 = (float)pi[a] / (float)sc.voxN[a], t1 = (float)(pi[a] + 1) / (float)sc.voxN[a];
        vMin[a] = (1 - t0) * sc.wbMin[a] + t0 * sc.wbMax[a];      // Bounds3::Lerp
        vMax[a] = (1 - t1) * sc.wbMin[a] + t1 * sc.wbMax[a];
    }
    const DevLight light = sc.lights[j];
    float acc = 0.f;
    for (int i = 0; i < 128; ++i) {
        DevIt ref;
        ref.p = vec3((1 - ri[i]) * vMin[0] + ri[i] * vMax[0], (1 - ri[128 + i]) * vMin[1] + ri[128 + i] * vMax[1], (1 - ri[256 + i]) * vMin[2] + ri[256 + i] * vMax[2]);
        ref.pErr = vec3(); ref.n = vec3();
        vec3 wi; float pdf = 0.f; DevIt pl;
        const rgb Li = light_sample<true>(sc, light, ref, ri[384 + i], ri[512 + i], &wi, &pdf, &pl);
        if (pdf > 0) acc += luminance(Li) / pdf;
    }
    contrib[g] = acc;
}
// k_voxel_dist: one thread per voxel: the minimum weight (:286-296) and the Distribution1D constructor (core/sampling.h:57-70)
__global__ __launch_bounds__(256) void k_voxel_dist(uint32_t nLights, uint32_t nVox, float *func, float *cdf, float *funcInt) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nVox) return;
    float *f = func + (size_t)v * nLights, *c = cdf + (size_t)v * (nLights + 1);
    float sum = 0.f;
    for (uint32_t i = 0; i < nLights; ++i) sum += f[i];
    const float avg = sum / (float)(128ull * nLights);
    const float minContrib = (avg > 0) ? (float)(.001 * (double)avg) : 1.f;
    for (uint32_t i = 0; i < nLights; ++i) f[i] = sel_max(f[i], minContrib);
    const int n = (int)nLights;
    c[0] = 0;
    for (int i = 1; i < n + 1; ++i) c[i] = c[i - 1] + f[i - 1] / n;
    const float fi = c[n];
    if (fi == 0) for (int i = 1; i < n + 1; ++i) c[i] = float(i) / float(n);
    else for (int i = 1; i < n + 1; ++i) c[i] /= fi;
    funcInt[v] = fi;
}

__global__ void k_fill_u32(uint32_t *p, uint32_t v, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
static inline uint32_t blocks_for(size_t n, uint32_t bs) { return (uint32_t)((n + bs - 1) / bs); }

// HPRT_WIDE_WALK=0 in the environment, or hprt_debug_wide_walk(0) at run time (diagnostics hook, not part of include/hprt.h): plain
// renders keep the binary walk
static int g_wideWalk = -1;
static bool WideWalkEnabled() {
    static const bool fromEnv = [] { const char *e = getenv("HPRT_WIDE_WALK"); return !(e && atoi(e) == 0); }();
    return g_wideWalk < 0 ? fromEnv : g_wideWalk != 0;
}
// HPRT_TRACE_PROFILE=1 in the environment, or hprt_debug_trace_profile_mode(1): the traversal kernels run their phase-profile
// variants (g_traceProf; bench.py reads the wide walk's records per ray from one such render)
static int g_traceProfileMode = -1;
static bool TraceProfileEnabled() {
    static const bool fromEnv = getenv("HPRT_TRACE_PROFILE") != nullptr;
    return g_traceProfileMode < 0 ? fromEnv : g_traceProfileMode != 0;
}
extern "C" __attribute__((visibility("default"))) int hprt_debug_trace_profile_mode(int on) { const int was = TraceProfileEnabled() ? 1 : 0; g_traceProfileMode = on; return was; }
// which walk a plain render of this scene takes: 1 the leaf-exact wide walk (k_walk4), 0 the binary walk (k_trace)
bool WideWalkInUse(const DevScene &sc) { return sc.wide != nullptr && WideWalkEnabled(); }
extern "C" __attribute__((visibility("default"))) int hprt_debug_wide_walk(int on) { const int was = WideWalkEnabled() ? 1 : 0; g_wideWalk = on; return was; }
void LaunchTrace(hipStream_t st, const DevScene &sc, bool anyHit, bool count, const uint32_t *queue, const uint32_t *countPtr,
                 uint32_t countImm, uint32_t gridItems, const RayStream &rays, const HitStream &hits, uint8_t *occ,
                 DevCounters *counters, uint32_t *workCounter, uint4 *rayStats) {
    if (gridItems == 0) return;
    (void)hipMemsetAsync(workCounter, 0, sizeof(uint32_t), st);
    // persistent waves: enough blocks to fill 256 CUs at this kernel's occupancy, never more than the rays need
    // (HPRT_TRACE_MAX_BLOCKS: test hook — a handful of waves working through many queue chunks each, tests/test_gpu_parity.py)
    static const uint32_t blockCap = [] { const char *e = getenv("HPRT_TRACE_MAX_BLOCKS"); return e ? (uint32_t)std::max(1, atoi(e)) : 0xffffffffu; }();
    // workgroups per CU: five (LDS stack 32 KB, or four by registers in the kernels with the quadric code); seven for the plain
    // any-hit kernel of triangle-only scenes (66 registers, 20 KB)
    static const uint32_t anyPerCu = [] { const char *e = getenv("HPRT_TRACE_ANY_PER_CU"); return e ? (uint32_t)std::min((int)(HPRT_DEEP_THREADS / (256u * HPRT_TRACE_BLOCK)), std::max(1, atoi(e))) : (uint32_t)HPRT_ANY_WAVES; }();      // (the deep-stack area is laid out for that many workgroups per CU)
    const bool profile = TraceProfileEnabled();
    const bool plain = !count && !profile;
    const bool hasQuad = sc.nSpheres != 0u;
    static const uint32_t closestPerCu = [] { const char *e = getenv("HPRT_TRACE_CLOSEST_PER_CU"); return e ? (uint32_t)std::min(7, std::max(1, atoi(e))) : (uint32_t)HPRT_CLOSEST_WAVES; }();
    // workgroups per CU = waves per SIMD the variant is compiled for (instanced scenes included: their LDS stacks are shorter)
    const uint32_t perCu = !plain ? 4u : hasQuad ? (anyHit ? (uint32_t)HPRT_QUAD_ANY_WAVES : (uint32_t)HPRT_QUAD_CLOSEST_WAVES) : (anyHit ? anyPerCu : closestPerCu);
    const uint32_t maxBlocks = std::min(256u * perCu, blockCap);
    static_assert(HPRT_DEEP_THREADS >= 256u * HPRT_TRACE_BLOCK * HPRT_ANY_WAVES && HPRT_DEEP_THREADS >= 256u * HPRT_TRACE_BLOCK * HPRT_CLOSEST_WAVES &&
                  HPRT_DEEP_THREADS >= 256u * HPRT_TRACE_BLOCK * HPRT_QUAD_ANY_WAVES, "the deep-stack area must cover the largest trace grid");
    dim3 grid(std::min(blocks_for(gridItems, HPRT_TRACE_BLOCK), maxBlocks)), block(HPRT_TRACE_BLOCK);
    // rays per queue-head atomic: large launches take 512 at a time, small ones keep every wave busy
    const uint32_t nWaves = grid.x * (HPRT_TRACE_BLOCK / 64);
    static const uint32_t chunkMax = [] { const char *e = getenv("HPRT_TRACE_CHUNK_MAX"); return e ? (uint32_t)atoi(e) : 512u; }();
    static const uint32_t chunkDiv = [] { const char *e = getenv("HPRT_TRACE_CHUNK_DIV"); return e ? (uint32_t)atoi(e) : 4u; }();
    uint32_t chunk = gridItems / (nWaves * chunkDiv);
    chunk = std::max(64u, std::min(chunkMax, chunk)) & ~63u;
    static const TraceTune tuneClosest = DefaultTraceTune(false), tuneAny = DefaultTraceTune(true);
    TraceTune tune = anyHit ? tuneAny : tuneClosest;
    // closest hits in large scenes run longer interior stretches between leaves (57 nodes per ray in the living room,
    // 20 in killeroo-simple): a longer pair phase pays there (+2-3 % on the living room and the atrium, -1 % on killeroo)
    static const bool tuneFromEnv = getenv("HPRT_TRACE_TUNE") != nullptr;
    if (!anyHit && !tuneFromEnv && sc.nPairs > 100000u) { tune.parkLimit = 32; tune.stepLimit = 14; tune.primMin = 12; }      // (round-2 sweep, tools/sweep_tune.sh: atrium +3.6 %, living room +-0)
    const bool inst = sc.nInstances != 0u, quad = sc.nSpheres != 0u;
    // two-level scenes: a primitive-phase iteration that enters an instance costs two dependent memory round trips (the primitive, then
    // the instance's transform), so fewer, fuller iterations pay (tools/sweep_inst_tune.sh on instanced-10m: closest +7.6 %, any hit +5 %)
    static const bool tuneAnyFromEnv = getenv("HPRT_TRACE_TUNE_ANY") != nullptr;
    if (inst && !anyHit && !tuneFromEnv) { tune.parkLimit = 32; tune.stepLimit = 10; tune.primMin = 12; }
    if (inst && anyHit && !tuneAnyFromEnv) { tune.refillBelow = 48; tune.parkLimit = 32; tune.stepLimit = 10; tune.primMin = 6; }
    // Plain renders of scenes that have the four-wide structure take the leaf-exact walk (k_walk4; HPRT_WIDE_WALK=0 or
    // hprt_debug_wide_walk(0) keep the binary walk: A/B runs and the tests that hold the two against each other)
    if (!count && sc.wide != nullptr && WideWalkEnabled()) {
        static const uint32_t wClosestPerCu = [] { const char *e = getenv("HPRT_WALK4_CLOSEST_PER_CU"); return e ? (uint32_t)std::min((int)HPRT_WALK4_CLOSEST_WAVES, std::max(1, atoi(e))) : (uint32_t)HPRT_WALK4_CLOSEST_WAVES; }();
        static const uint32_t wAnyPerCu = [] { const char *e = getenv("HPRT_WALK4_ANY_PER_CU"); return e ? (uint32_t)std::min((int)HPRT_WALK4_ANY_WAVES, std::max(1, atoi(e))) : (uint32_t)HPRT_WALK4_ANY_WAVES; }();
        static const TraceTune wTuneClosest = [] { TraceTune t{52, 32, 8, 4, 12}; if (const char *e = getenv("HPRT_WALK4_TUNE")) sscanf(e, "%d,%d,%d,%d,%d", &t.refillBelow, &t.parkLimit, &t.stepLimit, &t.sphereLimit, &t.primMin); return t; }();
        static const TraceTune wTuneAny = [] { TraceTune t{52, 28, 6, 4, 4}; if (const char *e = getenv("HPRT_WALK4_TUNE_ANY")) sscanf(e, "%d,%d,%d,%d,%d", &t.refillBelow, &t.parkLimit, &t.stepLimit, &t.sphereLimit, &t.primMin); return t; }();
        // workgroups per CU = waves per SIMD the variant is compiled for
        const uint32_t wPerCu = profile ? 4u : quad ? (anyHit ? (uint32_t)HPRT_QUAD_ANY_WAVES : (uint32_t)HPRT_QUAD_CLOSEST_WAVES) : inst ? (anyHit ? (uint32_t)HPRT_WALK4_INST_ANY_WAVES : (uint32_t)HPRT_WALK4_INST_CLOSEST_WAVES) : anyHit ? wAnyPerCu : wClosestPerCu;
        const uint32_t wBlocks = std::min(256u * wPerCu, blockCap);
        dim3 wGrid(std::min(blocks_for(gridItems, HPRT_TRACE_BLOCK), wBlocks));
        const uint32_t wWaves = wGrid.x * (HPRT_TRACE_BLOCK / 64);
        uint32_t wChunk = gridItems / (wWaves * chunkDiv);
        wChunk = std::max(64u, std::min(chunkMax, wChunk)) & ~63u;
        TraceTune wTune = anyHit ? wTuneAny : wTuneClosest;
        // two-level scenes: entering and leaving instances makes a leaf-phase iteration expensive, so fewer, fuller ones pay
        // (tools/sweep_walk4.sh on instanced-10m: closest +2.5 %, any hit +13 % over the one-level defaults)
        static const bool wTuneEnv = getenv("HPRT_WALK4_TUNE") != nullptr, wTuneAnyEnv = getenv("HPRT_WALK4_TUNE_ANY") != nullptr;
        if (inst && !anyHit && !wTuneEnv) wTune = TraceTune{52, 40, 14, 4, 16};
        if (inst && anyHit && !wTuneAnyEnv) wTune = TraceTune{48, 40, 14, 4, 8};
#define HPRT_WALK4_LAUNCH(A, P, I, Q) hipLaunchKernelGGL((k_walk4<A, P, I, Q>), wGrid, block, 0, st, sc, queue, countPtr, countImm, rays, hits, occ, workCounter, wChunk, wTune)
        // (the profiling variant exists with the quadric code only)
#define HPRT_WALK4_PICK(A) do { if (profile) { if (inst) HPRT_WALK4_LAUNCH(A, true, true, true); else HPRT_WALK4_LAUNCH(A, true, false, true); } \
                                else if (inst) { if (quad) HPRT_WALK4_LAUNCH(A, false, true, true); else HPRT_WALK4_LAUNCH(A, false, true, false); } \
                                else { if (quad) HPRT_WALK4_LAUNCH(A, false, false, true); else HPRT_WALK4_LAUNCH(A, false, false, false); } } while (0)
        if (anyHit) HPRT_WALK4_PICK(true); else HPRT_WALK4_PICK(false);
#undef HPRT_WALK4_PICK
#undef HPRT_WALK4_LAUNCH
        return;
    }
#define HPRT_TRACE_LAUNCH(A, M, I, Q) hipLaunchKernelGGL((k_trace<A, M, I, Q>), grid, block, 0, st, sc, queue, countPtr, countImm, rays, hits, occ, counters, rayStats, workCounter, chunk, tune)
    // (the profiling variant exists with the quadric code only)
#define HPRT_TRACE_PICK(A, M) do { if (inst) { if (quad || M == 2) HPRT_TRACE_LAUNCH(A, M, true, true); else HPRT_TRACE_LAUNCH(A, M, true, (M == 2)); } \
                                   else { if (quad || M == 2) HPRT_TRACE_LAUNCH(A, M, false, true); else HPRT_TRACE_LAUNCH(A, M, false, (M == 2)); } } while (0)
    if (anyHit) {
        if (count) HPRT_TRACE_PICK(true, 1); else if (profile) HPRT_TRACE_PICK(true, 2); else HPRT_TRACE_PICK(true, 0);
    } else {
        if (count) HPRT_TRACE_PICK(false, 1); else if (profile) HPRT_TRACE_PICK(false, 2); else HPRT_TRACE_PICK(false, 0);
    }
#undef HPRT_TRACE_PICK
#undef HPRT_TRACE_LAUNCH
}
#ifdef HPRT_SHADE_PROF
extern "C" __attribute__((visibility("default"))) int hprt_debug_shade_profile(unsigned long long out[24 + 48]) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_shadeProf), sizeof(unsigned long long) * 24) != hipSuccess) return -1;
    return hipMemcpyFromSymbol(out + 24, HIP_SYMBOL(g_shadeLanes), sizeof(unsigned long long) * 48) == hipSuccess ? 0 : -1;
}
#endif
// diagnostics hook (not part of include/hprt.h): read and optionally clear the phase profile
extern "C" __attribute__((visibility("default"))) int hprt_debug_trace_profile(unsigned long long out[32], int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_traceProf), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_traceProf), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
void LaunchGenerate(hipStream_t st, const DevScene &sc, const RenderParams &rp, const PathStream &out, uint32_t s0, uint32_t nSlots,
                    const IrregularSink *irr) {
    IrregularSink sink; memset(&sink, 0, sizeof(sink));
    if (irr) sink = *irr;
    if (nSlots) hipLaunchKernelGGL(k_generate, dim3(blocks_for(nSlots, 256)), dim3(256), 0, st, sc, rp, out, s0, nSlots, sink);
}
void LaunchBin(hipStream_t st, const DevScene &sc, const PathStream &in, const HitStream &hit, const uint32_t *queue,
               const uint32_t *countPtr, uint32_t countImm, uint32_t gridItems, int32_t maxDepth, int32_t bounces, const BinSet &bins,
               float4 *Lfinal) {
    if (gridItems) hipLaunchKernelGGL(k_bin, dim3(blocks_for(gridItems, HPRT_BIN_ITEMS * 1024)), dim3(1024), 0, st, sc, in, hit, queue, countPtr, countImm, maxDepth, bounces, bins, Lfinal);
}
void LaunchShade(hipStream_t st, int mode, const DevScene &sc, const RenderParams &rp, const PathStream &in, const HitStream &hit,
                 uint32_t gridItems, uint32_t s0, const PathStream &out, const VertexStreams &vs, const QueueSet &q,
                 const BinSet &bins, float4 *Lfinal, bool firstBounce, bool retryPass) {
    if (gridItems == 0) return;
    // workgroup size of the specialised variants (HPRT_SHADE_BLOCK = 1024 | 512 | 256); measured on
    // killeroo-simple: 512 is 3 % faster per frame than 1024 (two decoupled workgroups per CU instead of
    // one), 256 and forced higher occupancy (register spills) are slower
    static const int shadeCfg = [] {
        const char *e = getenv("HPRT_SHADE_BLOCK");
        const std::string v = e ? e : "512";
        return v == "1024" ? 0 : v == "256" ? 2 : 1;
    }();
    const bool specialised = mode == (int)BIN_MATTE || mode == (int)BIN_PLASTIC || mode == (int)BIN_SUBSTRATE;
    const uint32_t bs = !specialised ? 256u : (shadeCfg == 0 ? 1024u : shadeCfg == 1 ? 512u : 256u);
    dim3 grid(blocks_for(gridItems, bs)), block(bs);
#define HPRT_SHADE_LAUNCH_I(M, B, I) hipLaunchKernelGGL((k_shade<M, B, false, I>), grid, block, 0, st, sc, rp, in, hit, s0, out, vs, q, bins, Lfinal, firstBounce ? 1u : 0u, retryPass ? 1u : 0u)
#define HPRT_SHADE_LAUNCH(M, B) do { if (M != 2 && sc.nInstances != 0u) HPRT_SHADE_LAUNCH_I(M, B, (M != 2)); else HPRT_SHADE_LAUNCH_I(M, B, false); } while (0)
#define HPRT_SHADE_PICK(M) switch (shadeCfg) { case 0: HPRT_SHADE_LAUNCH(M, 1024); break; case 1: HPRT_SHADE_LAUNCH(M, 512); break; default: HPRT_SHADE_LAUNCH(M, 256); break; }
    if (mode == 3) hipLaunchKernelGGL((k_shade<2, 256, true>), grid, block, 0, st, sc, rp, in, hit, s0, out, vs, q, bins, Lfinal, firstBounce ? 1u : 0u, retryPass ? 1u : 0u);
    else if (mode == 2) { HPRT_SHADE_LAUNCH(2, 256); }
    else if (mode == 0) { HPRT_SHADE_PICK(0) }
    else if (mode == (int)BIN_SUBSTRATE) { HPRT_SHADE_PICK(3) }
    else { HPRT_SHADE_PICK(1) }
#undef HPRT_SHADE_PICK
#undef HPRT_SHADE_LAUNCH
}
void LaunchResolve(hipStream_t st, const DevScene &sc, const VertexStreams &vs, float4 *L, float4 *Lfinal, const uint32_t *queue,
                   const uint32_t *countPtr, uint32_t gridItems) {
    if (gridItems) hipLaunchKernelGGL(k_resolve, dim3(blocks_for(gridItems, 256)), dim3(256), 0, st, sc, vs, L, Lfinal, queue, countPtr);
}
void LaunchStoreRadiance(hipStream_t st, const float4 *Lfinal, float *LallR, float *LallG, float *LallB, uint32_t nPix, uint32_t s0,
                         uint32_t nSlots) {
    if (nSlots) hipLaunchKernelGGL(k_store_radiance, dim3(blocks_for(nSlots, 256)), dim3(256), 0, st, Lfinal, LallR, LallG, LallB, nPix, s0, nSlots);
}
__global__ __launch_bounds__(256) void k_pack_rays(const float *r7, uint32_t n, RayStream out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t N = n;
    out.a[i] = make_float4(r7[i], r7[N + i], r7[2 * N + i], r7[6 * N + i]);
    out.b[i] = make_float4(r7[3 * N + i], r7[4 * N + i], r7[5 * N + i], 0.f);
}
__global__ __launch_bounds__(256) void k_unpack_hits(HitStream h, uint32_t n, float *t, int32_t *prim, float *bary3) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a = h.a[i];
    t[i] = a.x; prim[i] = hit_prim(__float_as_int(a.y));
    if (bary3) { const size_t N = n; bary3[i] = a.z; bary3[N + i] = a.w; bary3[2 * N + i] = h.b[i].x; }
}
// Per-pixel GeneralStats (the fork's heat-map data, core/film.h:91 + core/integrator.cpp:327-328): every traced ray adds
// its counters to the pixel of the camera sample it belongs to.  ids[slot].w is the path id; pix: [6][nPix] =
// primitiveIntersections, ...P, leafNodeTraversals, ...P, bvhTreeNodeTraversals, ...P.
__global__ __launch_bounds__(256) void k_pixel_stats(const uint4 *rayStats, const float4 *ids, const uint32_t *queue, const uint32_t *countPtr,
                                                     uint32_t countImm, uint32_t nPix, int anyHit, uint32_t *pix) {
    const uint32_t n = countPtr ? *countPtr : countImm;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t slot = queue ? queue[i] : i;
    const uint4 c = rayStats[slot];
    const uint32_t p = (ids ? __float_as_uint(ids[slot].w) : slot) % nPix;      // ids == nullptr: camera rays, path id = slot
    uint32_t *base = pix + (anyHit ? nPix : 0u) + p;
    if (c.z) atomicAdd(base, c.z);
    if (c.y) atomicAdd(base + 2 * (size_t)nPix, c.y);
    if (c.x) atomicAdd(base + 4 * (size_t)nPix, c.x);
}
void LaunchPixelStats(hipStream_t st, const uint4 *rayStats, const float4 *ids, const uint32_t *queue, const uint32_t *countPtr,
                      uint32_t countImm, uint32_t gridItems, uint32_t nPix, bool anyHit, uint32_t *pix) {
    if (gridItems) hipLaunchKernelGGL(k_pixel_stats, dim3(blocks_for(gridItems, 256)), dim3(256), 0, st, rayStats, ids, queue, countPtr, countImm, nPix, anyHit ? 1 : 0, pix);
}
// local pixel -> film pixel: out[filmIndex][7] = {rays, the six counters}
__global__ __launch_bounds__(256) void k_pixel_stats_to_film(const uint32_t *pix, const uint32_t *pixelXY, uint32_t nPix, uint32_t spp, int cx0, int cy0,
                                                             int width, unsigned long long *out7) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nPix) return;
    const uint32_t pxy = pixelXY[p];
    const size_t f = (size_t)((int)(pxy >> 16) - cy0) * (size_t)width + (size_t)((int)(pxy & 0xffffu) - cx0);
    unsigned long long *o = out7 + 7 * f;
    o[0] = spp;
    for (int k = 0; k < 6; ++k) o[1 + k] = pix[(size_t)k * nPix + p];
}
void LaunchPixelStatsToFilm(hipStream_t st, const uint32_t *pix, const uint32_t *pixelXY, uint32_t nPix, uint32_t spp, int cx0, int cy0, int width,
                            unsigned long long *out7) {
    if (nPix) hipLaunchKernelGGL(k_pixel_stats_to_film, dim3(blocks_for(nPix, 256)), dim3(256), 0, st, pix, pixelXY, nPix, spp, cx0, cy0, width, out7);
}
__global__ __launch_bounds__(256) void k_capture_rays(const uint32_t *queue, uint32_t n, RayStream rays, float *out7, uint32_t cap) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = queue ? queue[i] : i;
    const float4 a = rays.a[s], b = rays.b[s];
    const size_t N = cap;
    out7[i] = a.x; out7[N + i] = a.y; out7[2 * N + i] = a.z; out7[3 * N + i] = b.x; out7[4 * N + i] = b.y; out7[5 * N + i] = b.z; out7[6 * N + i] = a.w;
}
void LaunchCaptureRays(hipStream_t st, const uint32_t *queue, uint32_t n, const RayStream &rays, float *out7, uint32_t cap) {
    n = std::min(n, cap);
    if (n) hipLaunchKernelGGL(k_capture_rays, dim3(blocks_for(n, 256)), dim3(256), 0, st, queue, n, rays, out7, cap);
}
// On-box HBM stream bandwidth (bench.py's roofline.peak_measured): a float4 copy, four 16-byte loads in
// flight per lane, 256 consecutive bytes per wave and request.
__global__ __launch_bounds__(256) void k_stream_copy(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
    const size_t base = (size_t)blockIdx.x * 1024u + threadIdx.x;
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) if (base + k * 256u < n) v[k] = src[base + k * 256u];
#pragma unroll
    for (int k = 0; k < 4; ++k) if (base + k * 256u < n) dst[base + k * 256u] = v[k];
}
void LaunchStreamCopy(hipStream_t st, const float4 *src, float4 *dst, size_t n) {
    if (n) hipLaunchKernelGGL(k_stream_copy, dim3((uint32_t)((n + 1023) / 1024)), dim3(256), 0, st, src, dst, n);
}
// On-box ceiling of the access pattern k_trace lives on (bench.py's roofline.gather): every lane fetches its own 64-byte record
// with four 16-byte loads, the next record depending on the one just read, records picked the way a traversal picks BVH nodes
// (a level of a complete binary tree uniformly, then a node of that level: the top of the tree stays in L1 / L2, the bottom does
// not).  Launch shape of k_trace<closest>: 256 threads, 24 KB of LDS, six workgroups per CU.  No arithmetic besides the pick.
__global__ __launch_bounds__(256, 6) void k_gather_probe(const uint4 *__restrict__ rec, uint32_t mask, uint32_t levels, int iters, uint32_t *out) {
    __shared__ uint32_t occupancyPad[6144];
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u, acc = 0;
    if (threadIdx.x == 999u) occupancyPad[0] = 1u;
    for (int it = 0; it < iters; ++it) {
        idx = idx * 1664525u + 1013904223u;
        const uint32_t l = (idx >> 3) % levels;
        const uint32_t r = ((1u << l) | ((idx >> 8) & ((1u << l) - 1u))) & mask;
        const uint4 a = rec[4 * r], b = rec[4 * r + 1], c = rec[4 * r + 2], d = rec[4 * r + 3];
        acc += a.x ^ b.y ^ c.z ^ d.w;
        idx ^= a.x;
    }
    if (acc == 0x12345678u) out[0] = acc + occupancyPad[threadIdx.x];
}
void LaunchGatherProbe(hipStream_t st, const uint4 *records, uint32_t log2Records, int itersPerLane, uint32_t blocks, uint32_t *sink) {
    hipLaunchKernelGGL(k_gather_probe, dim3(blocks), dim3(256), 0, st, records, (1u << log2Records) - 1u, log2Records, itersPerLane, sink);
}
void LaunchPackRays(hipStream_t st, const float *rays7, uint32_t n, const RayStream &out) {
    if (n) hipLaunchKernelGGL(k_pack_rays, dim3(blocks_for(n, 256)), dim3(256), 0, st, rays7, n, out);
}
void LaunchUnpackHits(hipStream_t st, const HitStream &hits, uint32_t n, float *t, int32_t *prim, float *bary3) {
    if (n) hipLaunchKernelGGL(k_unpack_hits, dim3(blocks_for(n, 256)), dim3(256), 0, st, hits, n, t, prim, bary3);
}
void LaunchFindIrregular(hipStream_t st, const DevScene &sc, const RenderParams &rp, const FilmGeom &fg, uint32_t spp, uint32_t *count,
                         uint32_t capacity, IrregularSample *out) {
    const size_t total = (size_t)rp.nPix * spp;
    if (total) hipLaunchKernelGGL(k_find_irregular, dim3(blocks_for(total, 256)), dim3(256), 0, st, sc, rp, fg, spp, count, capacity, out);
}
void LaunchFilmOwn(hipStream_t st, const RenderParams &rp, const FilmGeom &fg, const float *LallR, const float *LallG, const float *LallB,
                   uint32_t spp, const FilmExtras &ex, float *film) {
    if (rp.nPix) hipLaunchKernelGGL(k_film_own, dim3(blocks_for(rp.nPix, 256)), dim3(256), 0, st, rp, fg, LallR, LallG, LallB, spp, ex, film);
}
void LaunchFilmForeign(hipStream_t st, const RenderParams &rp, const FilmGeom &fg, const float *LallR, const float *LallG,
                       const float *LallB, const FilmExtras &ex, float *film) {
    if (ex.nForeignDest) hipLaunchKernelGGL(k_film_foreign, dim3(blocks_for(ex.nForeignDest, 64)), dim3(64), 0, st, rp, fg, LallR, LallG, LallB, ex, film);
}
void LaunchVoxelDistributions(hipStream_t st, const DevScene &sc, const float *ri, uint32_t nVox, float *func, float *cdf, float *funcInt) {
    const size_t total = (size_t)nVox * sc.nLights;
    if (!total) return;
    hipLaunchKernelGGL(k_voxel_contrib, dim3(blocks_for(total, 256)), dim3(256), 0, st, sc, ri, nVox, func, (const uint32_t *)nullptr);
    hipLaunchKernelGGL(k_voxel_dist, dim3(blocks_for(nVox, 256)), dim3(256), 0, st, sc.nLights, nVox, func, cdf, funcInt);
}
__global__ __launch_bounds__(256) void k_voxel_assign(const uint32_t *voxList, uint32_t n, uint32_t rowBase, int32_t *voxSlot) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) voxSlot[voxList[k]] = (int32_t)(rowBase + k);
}
void LaunchVoxelFill(hipStream_t st, const DevScene &sc, const float *ri, uint32_t n, uint32_t rowBase, float *func, float *cdf, float *funcInt) {
    if (!n) return;
    const size_t total = (size_t)n * sc.nLights;
    hipLaunchKernelGGL(k_voxel_contrib, dim3(blocks_for(total, 256)), dim3(256), 0, st, sc, ri, n, func + (size_t)rowBase * sc.nLights, (const uint32_t *)sc.voxRequest);
    hipLaunchKernelGGL(k_voxel_dist, dim3(blocks_for(n, 256)), dim3(256), 0, st, sc.nLights, n, func + (size_t)rowBase * sc.nLights, cdf + (size_t)rowBase * (sc.nLights + 1), funcInt + rowBase);
    hipLaunchKernelGGL(k_voxel_assign, dim3(blocks_for(n, 256)), dim3(256), 0, st, (const uint32_t *)sc.voxRequest, n, rowBase, sc.voxSlot);
}
void LaunchFilmForeignExport(hipStream_t st, const RenderParams &rp, const FilmGeom &fg, const float *LallR, const float *LallG,
                             const float *LallB, const FilmExtras &ex, uint32_t nGroups, const uint32_t *groupDest, const uint32_t *groupTile,
                             FilmRecord *out) {
    if (nGroups) hipLaunchKernelGGL(k_film_foreign_export, dim3(blocks_for(nGroups, 64)), dim3(64), 0, st, rp, fg, LallR, LallG, LallB, ex, nGroups, groupDest, groupTile, out);
}
void LaunchFilmApplyRecords(hipStream_t st, const FilmRecord *rec, const uint32_t *destBegin, uint32_t nDest, float *film) {
    if (nDest) hipLaunchKernelGGL(k_film_apply_records, dim3(blocks_for(nDest, 64)), dim3(64), 0, st, rec, destBegin, nDest, film);
}

}  // namespace hprt
