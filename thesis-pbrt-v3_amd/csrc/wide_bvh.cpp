// hprt host side — collapse of the reference's binary BVH into four-wide records with outward-quantised child boxes
// (wide_bvh.h).  Nothing here changes which primitives a ray tests or in which order: the leaves, their exact boxes and
// the order of the reference's tree stay; only the interior culling becomes coarser (never finer) than the reference's.
#include "wide_bvh.h"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace hprt {
namespace {

inline bool IsLeaf(const BvhNode &n) { return (n.countAxis & 3u) == 3u; }
inline float Pow2(int biased) { uint32_t u = (uint32_t)biased << 23; float f; memcpy(&f, &u, 4); return f; }
// what the kernel computes for a grid coordinate: one fused multiply-add (q * step is exact, so this is also origin + q * step)
inline float Dequant(int q, float step, float origin) { return std::fmaf((float)q, step, origin); }

// One axis of one node: origin = smallest lower bound; the smallest step 2^e for which every upper bound is reached with q <= 255;
// lower bounds rounded down, upper bounds rounded up ON THE DEQUANTISED VALUES.
bool QuantiseAxis(const float *lo, const float *hi, const bool *valid, float *origin, int *expo, uint8_t *qlo, uint8_t *qhi) {
    float o = 0.f, top = 0.f; bool any = false;
    for (int s = 0; s < 4; ++s) {
        if (!valid[s]) continue;
        if (!std::isfinite(lo[s]) || !std::isfinite(hi[s]) || lo[s] > hi[s]) return false;
        if (!any) { o = lo[s]; top = hi[s]; any = true; } else { o = std::min(o, lo[s]); top = std::max(top, hi[s]); }
    }
    if (o == 0.f) o = 0.f;      // (+0: the sign of a zero origin never shows in origin + q * step > 0, and q = 0 gives the origin itself)
    *origin = o;
    const double ext = (double)top - (double)o;
    int e = 1;
    if (ext > 0) { int k; std::frexp(ext / 255.0, &k); e = std::max(1, k + 126); }      // 2^(e - 127) >= ext / 255, roughly; corrected below
    for (; e <= 254; ++e) {
        const float step = Pow2(e);
        bool fits = true;
        for (int s = 0; s < 4 && fits; ++s) {
            if (!valid[s]) { qlo[s] = 0; qhi[s] = 0; continue; }
            double qd = std::floor(((double)lo[s] - (double)o) / (double)step);
            int q = (int)std::min(255.0, std::max(0.0, qd));
            while (q > 0 && Dequant(q, step, o) > lo[s]) --q;
            if (Dequant(q, step, o) > lo[s]) { fits = false; break; }      // (q == 0: the origin itself, never above a lower bound)
            qlo[s] = (uint8_t)q;
            qd = std::ceil(((double)hi[s] - (double)o) / (double)step);
            if (qd > 255.0) { fits = false; break; }
            q = (int)std::max(0.0, qd);
            while (q <= 255 && Dequant(q, step, o) < hi[s]) ++q;
            if (q > 255) { fits = false; break; }
            qhi[s] = (uint8_t)q;
        }
        if (fits) { *expo = e; return true; }
    }
    return false;
}

}  // namespace

bool BuildWide(const BvhNode *nd, uint32_t nNodes, const int32_t *leafRef, std::vector<DevWide> *out, int *stackNeed) {
    *stackNeed = 0;
    if (nNodes == 0) return true;
    const size_t first = out->size();
    struct Todo { uint32_t node; size_t wide; };
    std::vector<Todo> todo;
    auto blank = [] { DevWide w; memset(&w, 0, sizeof(w)); for (int s = 0; s < 4; ++s) w.ref[s] = WIDE_NONE; return w; };
    auto finish = [&](DevWide &w, const uint32_t child[4], const bool valid[4]) -> bool {
        float lo[3][4], hi[3][4];
        for (int s = 0; s < 4; ++s)
            for (int a = 0; a < 3; ++a) { lo[a][s] = valid[s] ? nd[child[s]].bmin[a] : 0.f; hi[a][s] = valid[s] ? nd[child[s]].bmax[a] : 0.f; }
        for (int a = 0; a < 3; ++a) {
            int e; uint8_t ql[4], qh[4];
            if (!QuantiseAxis(lo[a], hi[a], valid, &w.o[a], &e, ql, qh)) return false;
            w.em |= (uint32_t)e << (8 * a);
            w.q[2 * a] = (uint32_t)ql[0] | (uint32_t)ql[1] << 8 | (uint32_t)ql[2] << 16 | (uint32_t)ql[3] << 24;
            w.q[2 * a + 1] = (uint32_t)qh[0] | (uint32_t)qh[1] << 8 | (uint32_t)qh[2] << 16 | (uint32_t)qh[3] << 24;
        }
        return true;
    };
    if (IsLeaf(nd[0])) {
        // a one-leaf aggregate: a record whose only slot is that leaf (its box test stays the leaf's own, in the kernel)
        DevWide w = blank();
        const uint32_t child[4] = {0u, 0u, 0u, 0u}; const bool valid[4] = {true, false, false, false};
        w.ref[0] = leafRef[0];
        if (!finish(w, child, valid)) return false;
        out->push_back(w);
        return true;
    }
    out->push_back(blank());
    todo.push_back({0u, first});
    while (!todo.empty()) {
        const Todo t = todo.back(); todo.pop_back();
        const BvhNode &n = nd[t.node];
        const uint32_t c[2] = {t.node + 1u, (uint32_t)n.offset};
        uint32_t child[4] = {0u, 0u, 0u, 0u}; bool valid[4] = {false, false, false, false};
        uint32_t meta = n.countAxis & 3u;
        for (int g = 0; g < 2; ++g) {
            const BvhNode &k = nd[c[g]];
            if (IsLeaf(k)) { child[2 * g] = c[g]; valid[2 * g] = true; }
            else {
                child[2 * g] = c[g] + 1u; child[2 * g + 1] = (uint32_t)k.offset; valid[2 * g] = valid[2 * g + 1] = true;
                meta |= (k.countAxis & 3u) << (2 + 2 * g);
            }
        }
        DevWide w = blank();
        w.em = meta << 24;
        // (children are allocated together, the last one first on the to-do list so that the first child's subtree follows its parent)
        size_t idx[4];
        for (int s = 0; s < 4; ++s)
            if (valid[s] && !IsLeaf(nd[child[s]])) { idx[s] = out->size(); out->push_back(blank()); }
        for (int s = 3; s >= 0; --s) {
            if (!valid[s]) continue;
            if (IsLeaf(nd[child[s]])) w.ref[s] = leafRef[child[s]];
            else { if (idx[s] > 0x7ffffff0ull) return false; w.ref[s] = (int32_t)idx[s]; todo.push_back({child[s], idx[s]}); }
        }
        if (!finish(w, child, valid)) return false;
        (*out)[t.wide] = w;
    }
    // pending entries: a record with k hit slots leaves k - 1 on the stack while its first one is walked (children have larger indices)
    std::vector<int> need(out->size() - first, 0);
    for (size_t i = out->size(); i-- > first;) {
        const DevWide &w = (*out)[i];
        int k = 0, deepest = 0;
        for (int s = 0; s < 4; ++s) {
            if (w.ref[s] == WIDE_NONE) continue;
            ++k;
            if (w.ref[s] >= 0) deepest = std::max(deepest, need[(size_t)w.ref[s] - first]);
        }
        need[i - first] = std::max(0, k - 1) + deepest;
    }
    *stackNeed = need[0];
    return true;
}

}  // namespace hprt
