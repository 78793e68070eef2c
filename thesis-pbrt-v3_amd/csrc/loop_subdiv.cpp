// hprt host side — Loop subdivision to a limit-surface triangle mesh with normals.
// Behaviour follows the pbrt-v3 fork's shapes/loopsubdiv.cpp:150-397 (same stencil
// weights, same accumulation order, same one-ring walk), restated on index arrays
// instead of arena pointers.  Where the reference orders the two ends of an edge by
// pointer value (SDEdge ctor, :97-102) this uses creation order; the two agree for
// level-0 vertices (one contiguous array), i.e. exactly for "nlevels 1" meshes such
// as the bundled killeroo; deeper levels depend on the reference's allocator
// addresses and are documented as unpinned in DESIGN.md.
#include <cmath>
#include <map>
#include <string>
#include <vector>
#include "hprt_math.h"

namespace hprt {
namespace {

inline int nxt(int i) { return (i + 1) % 3; }
inline int prv(int i) { return (i + 2) % 3; }

struct SVert { vec3 p; int startFace = -1; int child = -1; bool regular = false, boundary = false; };
struct SFace {
    int v[3] = {-1, -1, -1}, f[3] = {-1, -1, -1}, kids[4] = {-1, -1, -1, -1};
    int vnum(int vert) const { for (int i = 0; i < 3; ++i) if (v[i] == vert) return i; return -1; }
};
struct Level {
    std::vector<SVert> V; std::vector<SFace> F;
    int nextFace(int face, int vert) const { return F[face].f[F[face].vnum(vert)]; }
    int prevFace(int face, int vert) const { return F[face].f[prv(F[face].vnum(vert))]; }
    int nextVert(int face, int vert) const { return F[face].v[nxt(F[face].vnum(vert))]; }
    int prevVert(int face, int vert) const { return F[face].v[prv(F[face].vnum(vert))]; }
    int otherVert(int face, int v0, int v1) const {
        for (int i = 0; i < 3; ++i) if (F[face].v[i] != v0 && F[face].v[i] != v1) return F[face].v[i];
        return -1;
    }
    int valence(int vi) const {                       // :128-143
        const SVert &vt = V[vi];
        int f = vt.startFace;
        if (!vt.boundary) {
            int nf = 1;
            while ((f = nextFace(f, vi)) != vt.startFace) ++nf;
            return nf;
        } else {
            int nf = 1;
            while ((f = nextFace(f, vi)) != -1) ++nf;
            f = vt.startFace;
            while ((f = prevFace(f, vi)) != -1) ++nf;
            return nf + 1;
        }
    }
    void oneRing(int vi, std::vector<vec3> &ring) const {   // :433-453
        ring.clear();
        const SVert &vt = V[vi];
        if (!vt.boundary) {
            int face = vt.startFace;
            do {
                ring.push_back(V[nextVert(face, vi)].p);
                face = nextFace(face, vi);
            } while (face != vt.startFace);
        } else {
            int face = vt.startFace, f2;
            while ((f2 = nextFace(face, vi)) != -1) face = f2;
            ring.push_back(V[nextVert(face, vi)].p);
            do {
                ring.push_back(V[prevVert(face, vi)].p);
                face = prevFace(face, vi);
            } while (face != -1);
        }
    }
    vec3 weightOneRing(int vi, float beta) const {    // :422-431
        std::vector<vec3> ring;
        oneRing(vi, ring);
        int val = (int)ring.size();
        vec3 p = (1 - val * beta) * V[vi].p;
        for (int i = 0; i < val; ++i) p = p + beta * ring[i];
        return p;
    }
    vec3 weightBoundary(int vi, float beta) const {   // :455-465
        std::vector<vec3> ring;
        oneRing(vi, ring);
        int val = (int)ring.size();
        vec3 p = (1 - 2 * beta) * V[vi].p;
        p = p + beta * ring[0];
        p = p + beta * ring[val - 1];
        return p;
    }
};
inline float betaW(int valence) { return valence == 3 ? 3.f / 16.f : 3.f / (8.f * valence); }   // :145-150
inline float loopGamma(int valence) { return 1.f / (valence + 3.f / (8.f * betaW(valence))); }  // :152-154

typedef std::pair<int, int> EdgeKey;
inline EdgeKey edgeKey(int a, int b) { return a < b ? EdgeKey(a, b) : EdgeKey(b, a); }

}  // namespace

bool LoopSubdivide(int nLevels, const std::vector<int> &indices, const std::vector<float> &P,
                   std::vector<int> *outIdx, std::vector<float> *outP, std::vector<float> *outN, std::string *err) {
    int nVertices = (int)(P.size() / 3), nFaces = (int)(indices.size() / 3);
    Level cur;
    cur.V.resize(nVertices); cur.F.resize(nFaces);
    for (int i = 0; i < nVertices; ++i) cur.V[i].p = vec3(P[3 * i], P[3 * i + 1], P[3 * i + 2]);
    for (int i = 0; i < nFaces; ++i)
        for (int j = 0; j < 3; ++j) {
            int v = indices[3 * i + j];
            if (v < 0 || v >= nVertices) { *err = "loopsubdiv: vertex index out of range"; return false; }
            cur.F[i].v[j] = v;
            cur.V[v].startFace = i;
        }
    // neighbour pointers (:178-197)
    {
        struct Half { int face, edgeNum; };
        std::map<EdgeKey, Half> open;
        for (int i = 0; i < nFaces; ++i)
            for (int e = 0; e < 3; ++e) {
                EdgeKey k = edgeKey(cur.F[i].v[e], cur.F[i].v[nxt(e)]);
                auto it = open.find(k);
                if (it == open.end()) open[k] = Half{i, e};
                else {
                    cur.F[it->second.face].f[it->second.edgeNum] = i;
                    cur.F[i].f[e] = it->second.face;
                    open.erase(it);
                }
            }
    }
    // boundary / regular flags (:199-212)
    for (int i = 0; i < nVertices; ++i) {
        SVert &v = cur.V[i];
        if (v.startFace < 0) { *err = "loopsubdiv: vertex not referenced by any face"; return false; }
        int f = v.startFace;
        do { f = cur.nextFace(f, i); } while (f != -1 && f != v.startFace);
        v.boundary = (f == -1);
        if (!v.boundary && cur.valence(i) == 6) v.regular = true;
        else if (v.boundary && cur.valence(i) == 4) v.regular = true;
        else v.regular = false;
    }
    // refinement (:214-305)
    for (int level = 0; level < nLevels; ++level) {
        Level nx;
        int nv = (int)cur.V.size(), nf = (int)cur.F.size();
        nx.V.resize(nv);
        for (int i = 0; i < nv; ++i) {
            cur.V[i].child = i;
            nx.V[i].regular = cur.V[i].regular; nx.V[i].boundary = cur.V[i].boundary;
        }
        nx.F.resize(4 * (size_t)nf);
        for (int i = 0; i < nf; ++i) for (int k = 0; k < 4; ++k) cur.F[i].kids[k] = 4 * i + k;
        // even vertices
        for (int i = 0; i < nv; ++i) {
            const SVert &v = cur.V[i];
            if (!v.boundary) {
                if (v.regular) nx.V[i].p = cur.weightOneRing(i, 1.f / 16.f);
                else nx.V[i].p = cur.weightOneRing(i, betaW(cur.valence(i)));
            } else nx.V[i].p = cur.weightBoundary(i, 1.f / 8.f);
        }
        // odd vertices
        std::map<EdgeKey, int> edgeVerts;
        for (int fi = 0; fi < nf; ++fi) {
            const SFace &face = cur.F[fi];
            for (int k = 0; k < 3; ++k) {
                EdgeKey ek = edgeKey(face.v[k], face.v[nxt(k)]);
                if (edgeVerts.count(ek)) continue;
                SVert nvtx;
                nvtx.regular = true;
                nvtx.boundary = (face.f[k] == -1);
                nvtx.startFace = face.kids[3];
                const vec3 &e0 = cur.V[ek.first].p, &e1 = cur.V[ek.second].p;
                if (nvtx.boundary) {
                    nvtx.p = 0.5f * e0;
                    nvtx.p = nvtx.p + 0.5f * e1;
                } else {
                    nvtx.p = 3.f / 8.f * e0;
                    nvtx.p = nvtx.p + 3.f / 8.f * e1;
                    nvtx.p = nvtx.p + 1.f / 8.f * cur.V[cur.otherVert(fi, ek.first, ek.second)].p;
                    nvtx.p = nvtx.p + 1.f / 8.f * cur.V[cur.otherVert(face.f[k], ek.first, ek.second)].p;
                }
                edgeVerts[ek] = (int)nx.V.size();
                nx.V.push_back(nvtx);
            }
        }
        // topology of the new level
        for (int i = 0; i < nv; ++i) {
            int sf = cur.V[i].startFace;
            nx.V[i].startFace = cur.F[sf].kids[cur.F[sf].vnum(i)];
        }
        for (int fi = 0; fi < nf; ++fi) {
            const SFace &face = cur.F[fi];
            for (int j = 0; j < 3; ++j) {
                nx.F[face.kids[3]].f[j] = face.kids[nxt(j)];
                nx.F[face.kids[j]].f[nxt(j)] = face.kids[3];
                int f2 = face.f[j];
                nx.F[face.kids[j]].f[j] = f2 != -1 ? cur.F[f2].kids[cur.F[f2].vnum(face.v[j])] : -1;
                f2 = face.f[prv(j)];
                nx.F[face.kids[j]].f[prv(j)] = f2 != -1 ? cur.F[f2].kids[cur.F[f2].vnum(face.v[j])] : -1;
            }
        }
        for (int fi = 0; fi < nf; ++fi) {
            const SFace &face = cur.F[fi];
            for (int j = 0; j < 3; ++j) {
                nx.F[face.kids[j]].v[j] = cur.V[face.v[j]].child;
                int vert = edgeVerts[edgeKey(face.v[j], face.v[nxt(j)])];
                nx.F[face.kids[j]].v[nxt(j)] = vert;
                nx.F[face.kids[nxt(j)]].v[j] = vert;
                nx.F[face.kids[3]].v[j] = vert;
            }
        }
        cur = std::move(nx);
    }
    // limit surface (:307-315)
    int nv = (int)cur.V.size();
    std::vector<vec3> pLimit(nv);
    for (int i = 0; i < nv; ++i) {
        if (cur.V[i].boundary) pLimit[i] = cur.weightBoundary(i, 1.f / 5.f);
        else pLimit[i] = cur.weightOneRing(i, loopGamma(cur.valence(i)));
    }
    for (int i = 0; i < nv; ++i) cur.V[i].p = pLimit[i];
    // tangents -> normals (:317-352); libm cosf/sinf as in the reference (load time only)
    outN->resize(3 * (size_t)nv);
    std::vector<vec3> ring;
    for (int i = 0; i < nv; ++i) {
        vec3 S(0, 0, 0), T(0, 0, 0);
        int valence = cur.valence(i);
        cur.oneRing(i, ring);
        if (!cur.V[i].boundary) {
            for (int j = 0; j < valence; ++j) {
                S = S + std::cos(2 * HPRT_PI * j / valence) * ring[j];
                T = T + std::sin(2 * HPRT_PI * j / valence) * ring[j];
            }
        } else {
            S = ring[valence - 1] - ring[0];
            if (valence == 2) T = ring[0] + ring[1] - 2 * cur.V[i].p;
            else if (valence == 3) T = ring[1] - cur.V[i].p;
            else if (valence == 4)
                T = -1 * ring[0] + 2 * ring[1] + 2 * ring[2] + -1 * ring[3] + -2 * cur.V[i].p;
            else {
                float theta = HPRT_PI / float(valence - 1);
                T = std::sin(theta) * (ring[0] + ring[valence - 1]);
                for (int k = 1; k < valence - 1; ++k) {
                    float wt = (2 * std::cos(theta) - 2) * std::sin((k)*theta);
                    T = T + wt * ring[k];
                }
                T = -T;
            }
        }
        vec3 n = cross(S, T);
        (*outN)[3 * i] = n.x; (*outN)[3 * i + 1] = n.y; (*outN)[3 * i + 2] = n.z;
    }
    outP->resize(3 * (size_t)nv);
    for (int i = 0; i < nv; ++i) { (*outP)[3 * i] = pLimit[i].x; (*outP)[3 * i + 1] = pLimit[i].y; (*outP)[3 * i + 2] = pLimit[i].z; }
    outIdx->resize(3 * cur.F.size());
    for (size_t i = 0; i < cur.F.size(); ++i) for (int j = 0; j < 3; ++j) (*outIdx)[3 * i + j] = cur.F[i].v[j];
    return true;
}

}  // namespace hprt
