// hprt — multi-GPU half of the C ABI (include/hprt.h): the film gather that follows a tile-sharded Render.
//
// The reference merges each worker's FilmTile into Film::pixels under a mutex (Film::MergeFilmTile,
// core/film.cpp:118-132).  Here every GPU holds the film of ITS tiles (hprt_render with tile_begin / tile_stride and
// HPRT_RENDER_EXPORT_FOREIGN) and the merge is one RCCL step over xGMI:
//   1. ncclReduce(sum) of the per-rank films onto the root.  The addends are disjoint — a pixel is non-zero on the one
//      rank that owns its tile — so the sum is exact whatever order the collective adds in.
//   2. The few box-filter contributions that cross a tile border (a sample whose Halton offset is exactly 0, core/film.h:
//      136-143) travel as 24-byte records: counts by ncclAllGather, records by a group of ncclSend / ncclRecv (peer ->
//      root, each over its own xGMI link).  The root sorts them by (destination pixel, source tile) and adds them in that
//      order, which is the order the single-GPU film adds them in: the N-GPU film equals the 1-GPU film bit for bit.
// One process per GPU uses hprt_comm_* + hprt_film_gather; a single process that drives several GPUs (how a pbrt-side
// adapter would, one HprtScene per GPU) uses hprt_film_gather_local.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/hprt.h"
#include "device_state.h"

using namespace hprt;

static_assert(sizeof(HprtFilmRecord) == sizeof(FilmRecord) && sizeof(FilmRecord) == 24, "film record layout");
static_assert(HPRT_COMM_ID_BYTES == sizeof(ncclUniqueId), "ncclUniqueId size");

#define NCCL_TRY(expr)                                                                                  \
    do {                                                                                                \
        ncclResult_t r__ = (expr);                                                                      \
        if (r__ != ncclSuccess)                                                                         \
            return SetError(HPRT_E_DEVICE, std::string(#expr) + ": " + ncclGetErrorString(r__));       \
    } while (0)

struct HprtComm {
    ncclComm_t comm = nullptr;
    int rank = 0, nRanks = 1, device = 0;
    DevBuf counts, staging, destBegin;
    uint32_t *hostCounts = nullptr;      // pinned
    ~HprtComm() {
        if (hostCounts) (void)hipHostFree(hostCounts);
        if (comm) (void)ncclCommDestroy(comm);
    }
};

namespace {

// Sort by (destination, source tile) and add on `device` into `film`; `scratch` receives the uploads.
int ApplyRecords(std::vector<FilmRecord> &rec, float *film, size_t nPixels, DevBuf &recBuf, DevBuf &beginBuf, hipStream_t st) {
    if (rec.empty()) return HPRT_OK;
    std::sort(rec.begin(), rec.end(), [](const FilmRecord &a, const FilmRecord &b) { return a.dest != b.dest ? a.dest < b.dest : a.srcTile < b.srcTile; });
    std::vector<uint32_t> destBegin;
    for (size_t i = 0; i < rec.size(); ++i) {
        if (rec[i].dest >= nPixels) return SetError(HPRT_E_INVALID, "film record outside the film");
        if (i == 0 || rec[i].dest != rec[i - 1].dest) destBegin.push_back((uint32_t)i);
    }
    const uint32_t nDest = (uint32_t)destBegin.size();
    destBegin.push_back((uint32_t)rec.size());
    HIP_TRY(recBuf.alloc(rec.size() * sizeof(FilmRecord)));
    HIP_TRY(beginBuf.alloc(destBegin.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpyAsync(recBuf.p, rec.data(), rec.size() * sizeof(FilmRecord), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(beginBuf.p, destBegin.data(), destBegin.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    LaunchFilmApplyRecords(st, recBuf.as<FilmRecord>(), beginBuf.as<uint32_t>(), nDest, film);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));      // the host vectors above must outlive the copies
    return HPRT_OK;
}

int CheckGatherable(const HprtScene *s, size_t nPixels) {
    if (!s->foreignExported) return SetError(HPRT_E_INVALID, "film gather: the scene's last render did not set HPRT_RENDER_EXPORT_FOREIGN");
    if (nPixels != s->filmPixels) return SetError(HPRT_E_INVALID, "film gather: pixel count differs from the last render's film");
    return HPRT_OK;
}

}  // namespace

extern "C" {

int hprt_comm_unique_id(uint8_t id[HPRT_COMM_ID_BYTES]) try {
    if (!id) return SetError(HPRT_E_INVALID, "hprt_comm_unique_id: null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SetError(HPRT_E_NO_DEVICE, "no HIP device available (hprt has no CPU fallback)");
    ncclUniqueId u;
    NCCL_TRY(ncclGetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }

int hprt_comm_create(const uint8_t id[HPRT_COMM_ID_BYTES], int rank, int n_ranks, int device, HprtComm **out) try {
    if (!id || !out) return SetError(HPRT_E_INVALID, "hprt_comm_create: null argument");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return SetError(HPRT_E_INVALID, "hprt_comm_create: rank outside [0, n_ranks)");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SetError(HPRT_E_NO_DEVICE, "no HIP device available (hprt has no CPU fallback)");
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= n) return SetError(HPRT_E_INVALID, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<HprtComm> c(new HprtComm());
    c->rank = rank; c->nRanks = n_ranks; c->device = device;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    NCCL_TRY(ncclCommInitRank(&c->comm, n_ranks, u, rank));
    HIP_TRY(c->counts.alloc(((size_t)n_ranks + 1) * sizeof(uint32_t)));
    HIP_TRY(hipHostMalloc((void **)&c->hostCounts, ((size_t)n_ranks + 1) * sizeof(uint32_t)));
    *out = c.release();
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }

int hprt_comm_info(const HprtComm *c, int *rank, int *n_ranks, int *device) try {
    if (!c) return SetError(HPRT_E_INVALID, "hprt_comm_info: null argument");
    // what the communicator itself reports, not what the caller passed in
    int r = -1, n = -1, d = -1;
    NCCL_TRY(ncclCommUserRank(c->comm, &r));
    NCCL_TRY(ncclCommCount(c->comm, &n));
    NCCL_TRY(ncclCommCuDevice(c->comm, &d));
    if (rank) *rank = r;
    if (n_ranks) *n_ranks = n;
    if (device) *device = d;
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }

void hprt_comm_destroy(HprtComm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    delete c;
}

// A rank whose own arguments or state are unusable must not return before the collectives: its peers would wait in them for
// ever.  Every rank therefore reaches the count all-gather, a failing one sends COUNT_FAILED instead of its record count, and
// all ranks return an error together; the one failure that can only happen later (the root's staging allocation) is agreed
// on by a one-word all-reduce before the grouped reduce / send / recv.  Inside the group the first error is kept and
// ncclGroupEnd is ALWAYS called: a group left open makes the communicator unusable.
static const uint32_t COUNT_FAILED = 0xffffffffu;

int hprt_film_gather(HprtComm *c, HprtScene *s, float *d_film_xyzw, size_t n_pixels, int root, void *stream) try {
    if (!c || !s) return SetError(HPRT_E_INVALID, "hprt_film_gather: null argument");      // (nothing to take part with)
    int localRc = HPRT_OK; std::string localMsg;
    auto fail = [&](int code, const std::string &m) { if (localRc == HPRT_OK) { localRc = code; localMsg = m; } };
    if (root < 0 || root >= c->nRanks) fail(HPRT_E_INVALID, "hprt_film_gather: root outside the communicator");
    if (s->device != c->device) fail(HPRT_E_INVALID, "hprt_film_gather: scene and communicator live on different devices");
    if (!s->foreignExported) fail(HPRT_E_INVALID, "film gather: the scene's last render did not set HPRT_RENDER_EXPORT_FOREIGN");
    else if (n_pixels != s->filmPixels) fail(HPRT_E_INVALID, "film gather: pixel count differs from the last render's film");
    // the film of the LAST render: the caller's buffer if that render was given one, else the scene's own
    float *film = d_film_xyzw ? d_film_xyzw : s->lastFilm;
    if (!film) fail(HPRT_E_INVALID, "hprt_film_gather: no film (render first)");
    else if (s->lastFilm && film != s->lastFilm) fail(HPRT_E_INVALID, "hprt_film_gather: d_film_xyzw is not the buffer the scene's last render wrote");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    SceneCall call(s, st);
    const int n = c->nRanks;
    // ---- record counts of every rank (COUNT_FAILED: that rank cannot go on) ----
    uint32_t *dCounts = c->counts.as<uint32_t>();
    c->hostCounts[n] = localRc == HPRT_OK ? s->nForeignRecords : COUNT_FAILED;
    HIP_TRY(hipMemcpyAsync(dCounts + n, c->hostCounts + n, sizeof(uint32_t), hipMemcpyHostToDevice, st));
    NCCL_TRY(ncclAllGather(dCounts + n, dCounts, 1, ncclUint32, c->comm, st));
    HIP_TRY(hipMemcpyAsync(c->hostCounts, dCounts, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (localRc != HPRT_OK) return SetError(localRc, localMsg);
    for (int r = 0; r < n; ++r)
        if (c->hostCounts[r] == COUNT_FAILED) return SetError(HPRT_E_INVALID, "hprt_film_gather: rank " + std::to_string(r) + " reported a failure; nothing was merged");
    size_t total = 0;
    std::vector<size_t> offset((size_t)n + 1, 0);
    for (int r = 0; r < n; ++r) { offset[r] = total; total += c->hostCounts[r]; }
    offset[n] = total;
    // ---- the root's staging area; every rank learns whether it exists ----
    uint32_t bad = 0u;
    if (c->rank == root && c->staging.alloc(std::max<size_t>(1, total) * sizeof(FilmRecord)) != hipSuccess) bad = 1u;
    c->hostCounts[n] = bad;
    HIP_TRY(hipMemcpyAsync(dCounts + n, c->hostCounts + n, sizeof(uint32_t), hipMemcpyHostToDevice, st));
    NCCL_TRY(ncclAllReduce(dCounts + n, dCounts + n, 1, ncclUint32, ncclMax, c->comm, st));
    HIP_TRY(hipMemcpyAsync(c->hostCounts + n, dCounts + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (c->hostCounts[n] != 0u) return SetError(HPRT_E_DEVICE, "hprt_film_gather: the root could not allocate the record staging area; nothing was merged");
    // ---- films: one reduce; records: peer -> root ----
    constexpr size_t kWords = sizeof(FilmRecord) / sizeof(uint32_t);
    ncclResult_t firstErr = ncclSuccess; const char *firstWhat = "";
    auto note = [&](ncclResult_t r, const char *what) { if (r != ncclSuccess && firstErr == ncclSuccess) { firstErr = r; firstWhat = what; } };
    // (the reduce on its own, the point-to-point transfers as one group after it: every rank issues the same two steps in the same
    // order on the same stream, and nothing rests on how a group that mixes a collective with send / recv is scheduled)
    // (an error is kept, not returned at once: this rank still issues its transfers, so that no peer waits for them)
    note(ncclReduce(film, film, 4 * n_pixels, ncclFloat, ncclSum, root, c->comm, st), "ncclReduce");
    note(ncclGroupStart(), "ncclGroupStart");
    if (c->rank == root) {
        for (int r = 0; r < n; ++r)
            if (r != root && c->hostCounts[r])
                note(ncclRecv(c->staging.as<FilmRecord>() + offset[r], c->hostCounts[r] * kWords, ncclUint32, r, c->comm, st), "ncclRecv");
    } else if (s->nForeignRecords)
        note(ncclSend(s->foreignRecords.p, s->nForeignRecords * kWords, ncclUint32, root, c->comm, st), "ncclSend");
    note(ncclGroupEnd(), "ncclGroupEnd");
    if (firstErr != ncclSuccess) return SetError(HPRT_E_DEVICE, std::string("hprt_film_gather: ") + firstWhat + ": " + ncclGetErrorString(firstErr));
    if (c->rank != root) { HIP_TRY(hipStreamSynchronize(st)); return HPRT_OK; }
    if (s->nForeignRecords)
        HIP_TRY(hipMemcpyAsync(c->staging.as<FilmRecord>() + offset[root], s->foreignRecords.p, s->nForeignRecords * sizeof(FilmRecord), hipMemcpyDeviceToDevice, st));
    std::vector<FilmRecord> rec(total);
    if (total) HIP_TRY(hipMemcpyAsync(rec.data(), c->staging.p, total * sizeof(FilmRecord), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return ApplyRecords(rec, film, n_pixels, c->staging, c->destBegin, st);
} catch (...) { return hprt::HandleException(); }

extern "C++" {
namespace {
// hprt_film_gather_local's communicators: one clique per device list, created on first use.  The map is never destroyed by
// the C++ runtime (static destruction may run after the HIP runtime is gone, and a DevBuf's hipFree or ncclCommDestroy would
// then touch a dead runtime): hprt_film_gather_local_shutdown releases it explicitly.
struct Clique {
    std::vector<ncclComm_t> comms; std::vector<int> devs; DevBuf recBuf, beginBuf; int rootDev = 0;
    ~Clique() {
        (void)hipSetDevice(rootDev);
        (void)recBuf.alloc(0); (void)beginBuf.alloc(0);
        for (ncclComm_t cm : comms) if (cm) (void)ncclCommDestroy(cm);
    }
};
std::mutex &CliqueMutex() { static std::mutex *m = new std::mutex(); return *m; }
std::map<std::vector<int>, std::unique_ptr<Clique>> &Cliques() { static auto *m = new std::map<std::vector<int>, std::unique_ptr<Clique>>(); return *m; }
}  // namespace
}  // extern "C++"

int hprt_film_gather_local(HprtScene *const *per_gpu, float *const *d_films, int n, size_t n_pixels, int root) try {
    if (!per_gpu || n < 1 || root < 0 || root >= n) return SetError(HPRT_E_INVALID, "hprt_film_gather_local: bad argument");
    std::vector<int> devs(n);
    std::vector<float *> films(n);
    // (one process: every check runs before the first collective call, so an early return strands nobody)
    for (int i = 0; i < n; ++i) {
        if (!per_gpu[i]) return SetError(HPRT_E_INVALID, "hprt_film_gather_local: null scene");
        int rc = CheckGatherable(per_gpu[i], n_pixels);
        if (rc != HPRT_OK) return rc;
        devs[i] = per_gpu[i]->device;
        films[i] = (d_films && d_films[i]) ? d_films[i] : per_gpu[i]->lastFilm;
        if (!films[i]) return SetError(HPRT_E_INVALID, "hprt_film_gather_local: a scene has no film (render first)");
        if (per_gpu[i]->lastFilm && films[i] != per_gpu[i]->lastFilm) return SetError(HPRT_E_INVALID, "hprt_film_gather_local: a film pointer is not the buffer that scene's last render wrote");
        for (int k = 0; k < i; ++k) if (devs[k] == devs[i]) return SetError(HPRT_E_INVALID, "hprt_film_gather_local: two scenes on one device (RCCL refuses duplicate GPUs)");
    }
    std::lock_guard<std::mutex> lock(CliqueMutex());
    std::unique_ptr<Clique> &cl = Cliques()[devs];
    if (!cl) {
        std::unique_ptr<Clique> fresh(new Clique());
        fresh->comms.assign(n, nullptr); fresh->devs = devs;
        NCCL_TRY(ncclCommInitAll(fresh->comms.data(), n, devs.data()));
        cl = std::move(fresh);
    }
    cl->rootDev = devs[root];
    ncclResult_t firstErr = ncclSuccess; hipError_t firstHip = hipSuccess;
    NCCL_TRY(ncclGroupStart());
    for (int i = 0; i < n; ++i) {      // (keep going on an error: the group must be closed whatever happened inside it)
        const hipError_t he = hipSetDevice(devs[i]);
        if (he != hipSuccess) { if (firstHip == hipSuccess) firstHip = he; continue; }
        const ncclResult_t r = ncclReduce(films[i], films[i], 4 * n_pixels, ncclFloat, ncclSum, root, cl->comms[i], nullptr);
        if (r != ncclSuccess && firstErr == ncclSuccess) firstErr = r;
    }
    const ncclResult_t ge = ncclGroupEnd();
    if (firstHip != hipSuccess) return SetError(HPRT_E_DEVICE, std::string("hprt_film_gather_local: hipSetDevice: ") + hipGetErrorString(firstHip));
    if (firstErr != ncclSuccess || ge != ncclSuccess) return SetError(HPRT_E_DEVICE, std::string("hprt_film_gather_local: ncclReduce group: ") + ncclGetErrorString(firstErr != ncclSuccess ? firstErr : ge));
    // one process sees every device's memory: the records need no collective
    std::vector<FilmRecord> rec;
    for (int i = 0; i < n; ++i) {
        HIP_TRY(hipSetDevice(devs[i]));
        HIP_TRY(hipDeviceSynchronize());
        const size_t k = per_gpu[i]->nForeignRecords, at = rec.size();
        if (!k) continue;
        rec.resize(at + k);
        HIP_TRY(hipMemcpy(rec.data() + at, per_gpu[i]->foreignRecords.p, k * sizeof(FilmRecord), hipMemcpyDeviceToHost));
    }
    HIP_TRY(hipSetDevice(devs[root]));
    return ApplyRecords(rec, films[root], n_pixels, cl->recBuf, cl->beginBuf, nullptr);
} catch (...) { return hprt::HandleException(); }

void hprt_film_gather_local_shutdown(void) {
    std::lock_guard<std::mutex> lock(CliqueMutex());
    Cliques().clear();
}

int hprt_film_records_read(HprtScene *s, HprtFilmRecord *out, size_t capacity, size_t *n_records) try {
    if (!s || !n_records) return SetError(HPRT_E_INVALID, "hprt_film_records_read: null argument");
    if (!s->foreignExported) return SetError(HPRT_E_INVALID, "hprt_film_records_read: the last render did not set HPRT_RENDER_EXPORT_FOREIGN");
    *n_records = s->nForeignRecords;
    if (!out || s->nForeignRecords == 0) return HPRT_OK;
    if (capacity < s->nForeignRecords) return SetError(HPRT_E_INVALID, "hprt_film_records_read: buffer too small");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, s->foreignRecords.p, (size_t)s->nForeignRecords * sizeof(FilmRecord), hipMemcpyDeviceToHost));
    return HPRT_OK;
} catch (...) { return hprt::HandleException(); }

}  // extern "C"
