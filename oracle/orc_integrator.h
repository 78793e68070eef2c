// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
// Camera, film, PathIntegrator::Li and the SamplerIntegrator::Render tile loop:
// cameras/perspective.cpp, core/camera.h, core/film.{h,cpp}, core/integrator.cpp,
// integrators/path.cpp, core/light.cpp, core/lightdistrib.cpp, core/sampling.h.
#pragma once
#include <array>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include "orc_sampler.h"
#include "orc_shading.h"

namespace orc {

// ---- 4x4 helpers for the camera (core/transform.cpp) ----------------------
inline M44 MatMul(const M44 &a, const M44 &b) {   // Matrix4x4::Mul, transform.h:83-91
    M44 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
    return r;
}
inline M44 MatIdentity() { M44 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = (i == j) ? 1.f : 0.f; return r; }
// core/transform.cpp:82-139 (Gauss-Jordan with full pivoting)
inline M44 MatInverse(const M44 &m) {
    int indxc[4], indxr[4];
    int ipiv[4] = {0, 0, 0, 0};
    Float minv[4][4];
    memcpy(minv, m.m, 4 * 4 * sizeof(Float));
    for (int i = 0; i < 4; i++) {
        int irow = 0, icol = 0;
        Float big = 0.f;
        for (int j = 0; j < 4; j++) {
            if (ipiv[j] != 1) {
                for (int k = 0; k < 4; k++) {
                    if (ipiv[k] == 0) {
                        if (std::abs(minv[j][k]) >= big) { big = Float(std::abs(minv[j][k])); irow = j; icol = k; }
                    }
                }
            }
        }
        ++ipiv[icol];
        if (irow != icol) for (int k = 0; k < 4; ++k) std::swap(minv[irow][k], minv[icol][k]);
        indxr[i] = irow; indxc[i] = icol;
        Float pivinv = 1. / minv[icol][icol];
        minv[icol][icol] = 1.;
        for (int j = 0; j < 4; j++) minv[icol][j] *= pivinv;
        for (int j = 0; j < 4; j++) {
            if (j != icol) {
                Float save = minv[j][icol];
                minv[j][icol] = 0;
                for (int k = 0; k < 4; k++) minv[j][k] -= minv[icol][k] * save;
            }
        }
    }
    for (int j = 3; j >= 0; j--) {
        if (indxr[j] != indxc[j]) for (int k = 0; k < 4; k++) std::swap(minv[k][indxr[j]], minv[k][indxc[j]]);
    }
    M44 r; memcpy(r.m, minv, sizeof(minv)); return r;
}
struct Xf { M44 m, mInv; };
inline Xf XfMul(const Xf &a, const Xf &b) { return Xf{MatMul(a.m, b.m), MatMul(b.mInv, a.mInv)}; }   // transform.cpp:251-253
inline Xf XfInverse(const Xf &a) { return Xf{a.mInv, a.m}; }
inline Xf XfTranslate(const V3 &d) {      // transform.cpp:141-147
    Xf t; t.m = MatIdentity(); t.mInv = MatIdentity();
    t.m.m[0][3] = d.x; t.m.m[1][3] = d.y; t.m.m[2][3] = d.z;
    t.mInv.m[0][3] = -d.x; t.mInv.m[1][3] = -d.y; t.mInv.m[2][3] = -d.z;
    return t;
}
inline Xf XfScale(Float x, Float y, Float z) {   // transform.cpp:149-153
    Xf t; t.m = MatIdentity(); t.mInv = MatIdentity();
    t.m.m[0][0] = x; t.m.m[1][1] = y; t.m.m[2][2] = z;
    t.mInv.m[0][0] = 1 / x; t.mInv.m[1][1] = 1 / y; t.mInv.m[2][2] = 1 / z;
    return t;
}
// transform.cpp:303-311.  std::tan(float) is glibc tanf (host-side, once).
inline Xf XfPerspective(Float fov, Float n, Float f) {
    M44 persp = MatIdentity();
    persp.m[2][2] = f / (f - n); persp.m[2][3] = -f * n / (f - n);
    persp.m[3][2] = 1; persp.m[3][3] = 0;
    Float invTanAng = 1 / std::tan(Radians(fov) / 2);
    Xf p{persp, MatInverse(persp)};
    return XfMul(XfScale(invTanAng, invTanAng, 1), p);
}

// ---- Camera (core/camera.h:84-115, cameras/perspective.cpp:45-144) --------
// RayDifferential's extra members (core/geometry.h:1205-1240)
struct RayDiff {
    bool has = false;
    V3 rxOrigin, ryOrigin, rxDirection, ryDirection;
    void Scale(const Ray &r, Float s) {      // ScaleDifferentials
        rxOrigin = r.o + (rxOrigin - r.o) * s;
        ryOrigin = r.o + (ryOrigin - r.o) * s;
        rxDirection = r.d + (rxDirection - r.d) * s;
        ryDirection = r.d + (ryDirection - r.d) * s;
    }
};
// SurfaceInteraction::ComputeDifferentials, core/interaction.cpp:103-149 (the (u,v) part; dpdx/dpdy feed nothing here)
inline bool SolveLinearSystem2x2(const Float A[2][2], const Float B[2], Float *x0, Float *x1) {   // core/transform.cpp:41-49
    Float det = A[0][0] * A[1][1] - A[0][1] * A[1][0];
    if (std::abs(det) < 1e-10f) return false;
    *x0 = (A[1][1] * B[0] - A[0][1] * B[1]) / det;
    *x1 = (A[0][0] * B[1] - A[1][0] * B[0]) / det;
    if (std::isnan(*x0) || std::isnan(*x1)) return false;
    return true;
}
inline void ComputeDifferentials(SurfaceInteraction *si, const Ray &ray, const RayDiff &rd) {
    si->dudx = si->dvdx = si->dudy = si->dvdy = 0;
    if (!rd.has) return;
    const V3 &n = si->n, &p = si->p;
    Float d = Dot(n, V3(p.x, p.y, p.z));
    Float tx = -(Dot(n, rd.rxOrigin) - d) / Dot(n, rd.rxDirection);
    if (std::isinf(tx) || std::isnan(tx)) return;
    V3 px = rd.rxOrigin + tx * rd.rxDirection;
    Float ty = -(Dot(n, rd.ryOrigin) - d) / Dot(n, rd.ryDirection);
    if (std::isinf(ty) || std::isnan(ty)) return;
    V3 py = rd.ryOrigin + ty * rd.ryDirection;
    int dim[2];
    if (std::abs(n.x) > std::abs(n.y) && std::abs(n.x) > std::abs(n.z)) { dim[0] = 1; dim[1] = 2; }
    else if (std::abs(n.y) > std::abs(n.z)) { dim[0] = 0; dim[1] = 2; }
    else { dim[0] = 0; dim[1] = 1; }
    Float A[2][2] = {{si->dpdu[dim[0]], si->dpdv[dim[0]]}, {si->dpdu[dim[1]], si->dpdv[dim[1]]}};
    Float Bx[2] = {px[dim[0]] - p[dim[0]], px[dim[1]] - p[dim[1]]};
    Float By[2] = {py[dim[0]] - p[dim[0]], py[dim[1]] - p[dim[1]]};
    if (!SolveLinearSystem2x2(A, Bx, &si->dudx, &si->dvdx)) si->dudx = si->dvdx = 0;
    if (!SolveLinearSystem2x2(A, By, &si->dudy, &si->dvdy)) si->dudy = si->dvdy = 0;
    (void)ray;
}

struct Camera {
    Xf rasterToCamera;
    M44 camToWorld;
    Float lensRadius, focalDistance;
    V3 dxCamera, dyCamera;           // cameras/perspective.cpp:55-58
    void Init(const SceneParams &p) {
        Xf camToScreen = XfPerspective(p.fov, 1e-2f, 1000.f);
        const Float *sw = p.screenWindow;   // pMin.x pMax.x pMin.y pMax.y
        Xf screenToRaster = XfMul(XfMul(XfScale((Float)p.xres, (Float)p.yres, 1),
                                        XfScale(1 / (sw[1] - sw[0]), 1 / (sw[2] - sw[3]), 1)),
                                  XfTranslate(V3(-sw[0], -sw[3], 0)));
        Xf rasterToScreen = XfInverse(screenToRaster);
        rasterToCamera = XfMul(XfInverse(camToScreen), rasterToScreen);
        camToWorld = p.camToWorld;
        lensRadius = p.lensRadius; focalDistance = p.focalDistance;
        dxCamera = XfPoint(rasterToCamera.m, V3(1, 0, 0)) - XfPoint(rasterToCamera.m, V3(0, 0, 0));
        dyCamera = XfPoint(rasterToCamera.m, V3(0, 1, 0)) - XfPoint(rasterToCamera.m, V3(0, 0, 0));
    }
    // GenerateRayDifferential :95-144.  The differentials only feed texture filtering; rd (may be null) receives them.
    Float GenerateRay(const P2 &pFilmS, const P2 &pLensS, Ray *ray, RayDiff *rd = nullptr) const {
        V3 pFilm(pFilmS.x, pFilmS.y, 0);
        V3 pCamera = XfPoint(rasterToCamera.m, pFilm);
        V3 dir = Normalize(V3(pCamera.x, pCamera.y, pCamera.z));
        Ray r(V3(0, 0, 0), dir);
        if (lensRadius > 0) {
            P2 c = ConcentricSampleDisk(pLensS);
            P2 pLens(lensRadius * c.x, lensRadius * c.y);
            Float ft = focalDistance / r.d.z;
            V3 pFocus = r(ft);
            r.o = V3(pLens.x, pLens.y, 0);
            r.d = Normalize(pFocus - r.o);
        }
        // CameraToWorld(ray): Transform::operator()(const Ray&), transform.h:245-259
        V3 oError;
        V3 o = XfPointErr(camToWorld, r.o, &oError);
        V3 d = XfVector(camToWorld, r.d);
        Float lengthSquared = d.LengthSquared();
        Float tMax = r.tMax;
        if (lengthSquared > 0) {
            Float dt = Dot(Abs(d), oError) / lengthSquared;
            o += d * dt;
            tMax -= dt;
        }
        *ray = Ray(o, d, tMax);
        if (rd) {
            // offset rays (:117-138), then Transform::operator()(const RayDifferential &) (core/transform.h:266-275):
            // points and vectors transformed plainly
            V3 rxO, ryO, rxD, ryD;
            if (lensRadius > 0) {
                P2 c = ConcentricSampleDisk(pLensS);
                P2 pLens(lensRadius * c.x, lensRadius * c.y);
                V3 dx = Normalize(pCamera + dxCamera);
                Float ft = focalDistance / dx.z;
                V3 pFocus = V3(0, 0, 0) + (ft * dx);
                rxO = V3(pLens.x, pLens.y, 0);
                rxD = Normalize(pFocus - rxO);
                V3 dy = Normalize(pCamera + dyCamera);
                ft = focalDistance / dy.z;
                pFocus = V3(0, 0, 0) + (ft * dy);
                ryO = V3(pLens.x, pLens.y, 0);
                ryD = Normalize(pFocus - ryO);
            } else {
                rxO = ryO = r.o;
                rxD = Normalize(pCamera + dxCamera);
                ryD = Normalize(pCamera + dyCamera);
            }
            rd->has = true;
            rd->rxOrigin = XfPoint(camToWorld, rxO); rd->ryOrigin = XfPoint(camToWorld, ryO);
            rd->rxDirection = XfVector(camToWorld, rxD); rd->ryDirection = XfVector(camToWorld, ryD);
        }
        return 1;
    }
};

// ---- Film (core/film.{h,cpp}; box filter) ----------------------------------
struct FilmPixel { Float xyz[3] = {0, 0, 0}; Float filterWeightSum = 0; };
struct FilmTilePixel { Spec contribSum = 0.f; Float filterWeightSum = 0.f; };
struct Film {
    int xres, yres;
    int cx0, cy0, cx1, cy1;          // croppedPixelBounds
    Float radius[2], scale, maxSampleLuminance;
    static const int filterTableWidth = 16;
    Float filterTable[filterTableWidth * filterTableWidth];
    std::vector<FilmPixel> pixels;
    // Pixel::stats (core/film.h:91), the fork's per-pixel GeneralStats (core/geometry.h:1078-1173), BVH fields:
    // rays, primitiveIntersections, primitiveIntersectionsP, leafNodeTraversals, leafNodeTraversalsP,
    // bvhTreeNodeTraversals, bvhTreeNodeTraversalsP
    std::vector<std::array<uint64_t, 7>> stats;
    std::mutex mutex;
    void Init(const SceneParams &p) {
        xres = p.xres; yres = p.yres;
        cx0 = (int)std::ceil(xres * p.crop[0]); cx1 = (int)std::ceil(xres * p.crop[1]);   // film.cpp:56-60
        cy0 = (int)std::ceil(yres * p.crop[2]); cy1 = (int)std::ceil(yres * p.crop[3]);
        radius[0] = p.filterRadius[0]; radius[1] = p.filterRadius[1];
        scale = p.filmScale; maxSampleLuminance = p.maxSampleLuminance;
        for (int i = 0; i < filterTableWidth * filterTableWidth; ++i) filterTable[i] = 1.f;   // BoxFilter::Evaluate
        pixels.assign((size_t)smax(0, (cx1 - cx0) * (cy1 - cy0)), FilmPixel());
        stats.assign(pixels.size(), std::array<uint64_t, 7>{{0, 0, 0, 0, 0, 0, 0}});
    }
    void GetSampleBounds(int *x0, int *y0, int *x1, int *y1) const {   // film.cpp:81-87
        *x0 = (int)std::floor((Float)cx0 + 0.5f - radius[0]);
        *y0 = (int)std::floor((Float)cy0 + 0.5f - radius[1]);
        *x1 = (int)std::ceil((Float)cx1 - 0.5f + radius[0]);
        *y1 = (int)std::ceil((Float)cy1 - 0.5f + radius[1]);
    }
    FilmPixel &GetPixel(int x, int y) { return pixels[(size_t)(x - cx0) + (size_t)(y - cy0) * (cx1 - cx0)]; }
};
struct FilmTile {
    int bx0, by0, bx1, by1;
    const Film *film;
    std::vector<FilmTilePixel> pixels;
    FilmTile(const Film *f, int sx0, int sy0, int sx1, int sy1) : film(f) {   // film.cpp:96-107
        int p0x = (int)std::ceil((Float)sx0 - 0.5f - f->radius[0]);
        int p0y = (int)std::ceil((Float)sy0 - 0.5f - f->radius[1]);
        int p1x = (int)std::floor((Float)sx1 - 0.5f + f->radius[0]) + 1;
        int p1y = (int)std::floor((Float)sy1 - 0.5f + f->radius[1]) + 1;
        bx0 = smax(p0x, f->cx0); by0 = smax(p0y, f->cy0);
        bx1 = smin(p1x, f->cx1); by1 = smin(p1y, f->cy1);
        pixels.assign((size_t)smax(0, (bx1 - bx0) * (by1 - by0)), FilmTilePixel());
    }
    FilmTilePixel &GetPixel(int x, int y) { return pixels[(size_t)(x - bx0) + (size_t)(y - by0) * (bx1 - bx0)]; }
    void AddSample(const P2 &pFilm, Spec L, Float sampleWeight) {   // film.h:130-170
        if (L.y() > film->maxSampleLuminance) L *= film->maxSampleLuminance / L.y();
        P2 pFilmDiscrete(pFilm.x - 0.5f, pFilm.y - 0.5f);
        int p0x = (int)std::ceil(pFilmDiscrete.x - film->radius[0]);
        int p0y = (int)std::ceil(pFilmDiscrete.y - film->radius[1]);
        int p1x = (int)std::floor(pFilmDiscrete.x + film->radius[0]) + 1;
        int p1y = (int)std::floor(pFilmDiscrete.y + film->radius[1]) + 1;
        p0x = smax(p0x, bx0); p0y = smax(p0y, by0);
        p1x = smin(p1x, bx1); p1y = smin(p1y, by1);
        const int W = Film::filterTableWidth;
        Float invRx = 1 / film->radius[0], invRy = 1 / film->radius[1];
        for (int y = p0y; y < p1y; ++y) {
            Float fy = std::abs((y - pFilmDiscrete.y) * invRy * W);
            int iy = smin((int)std::floor(fy), W - 1);
            for (int x = p0x; x < p1x; ++x) {
                Float fx = std::abs((x - pFilmDiscrete.x) * invRx * W);
                int ix = smin((int)std::floor(fx), W - 1);
                Float filterWeight = film->filterTable[iy * W + ix];
                FilmTilePixel &pixel = GetPixel(x, y);
                pixel.contribSum += L * sampleWeight * filterWeight;
                pixel.filterWeightSum += filterWeight;
            }
        }
    }
};
inline void MergeFilmTile(Film *film, FilmTile &tile) {   // film.cpp:118-132
    std::lock_guard<std::mutex> lock(film->mutex);
    for (int y = tile.by0; y < tile.by1; ++y)
        for (int x = tile.bx0; x < tile.bx1; ++x) {
            const FilmTilePixel &tp = tile.GetPixel(x, y);
            FilmPixel &mp = film->GetPixel(x, y);
            Float xyz[3];
            RGBToXYZ(tp.contribSum.c, xyz);
            for (int i = 0; i < 3; ++i) mp.xyz[i] += xyz[i];
            mp.filterWeightSum += tp.filterWeightSum;
        }
}
// Film::WriteImage arithmetic, film.cpp:266-303 (no splats); rgb is top-to-bottom
inline void FilmToRGB(Film *film, std::vector<Float> *rgb) {
    size_t n = film->pixels.size();
    rgb->assign(3 * n, 0.f);
    for (size_t i = 0; i < n; ++i) {
        FilmPixel &pixel = film->pixels[i];
        Float *o = &(*rgb)[3 * i];
        XYZToRGB(pixel.xyz, o);
        Float filterWeightSum = pixel.filterWeightSum;
        if (filterWeightSum != 0) {
            Float invWt = (Float)1 / filterWeightSum;
            o[0] = smax((Float)0, o[0] * invWt);
            o[1] = smax((Float)0, o[1] * invWt);
            o[2] = smax((Float)0, o[2] * invWt);
        }
        Float splatRGB[3]; Float splatXYZ[3] = {0, 0, 0};
        XYZToRGB(splatXYZ, splatRGB);
        o[0] += 1.f * splatRGB[0]; o[1] += 1.f * splatRGB[1]; o[2] += 1.f * splatRGB[2];
        o[0] *= film->scale; o[1] *= film->scale; o[2] *= film->scale;
    }
}

// core/sampling.h:55-109 (uniform light distribution, lightdistrib.cpp:68-75)
struct Distribution1D {
    std::vector<Float> func, cdf; Float funcInt;
    void Init(const Float *f, int n) {
        func.assign(f, f + n); cdf.assign(n + 1, 0.f);
        cdf[0] = 0;
        for (int i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / n;
        funcInt = cdf[n];
        if (funcInt == 0) for (int i = 1; i < n + 1; ++i) cdf[i] = Float(i) / Float(n);
        else for (int i = 1; i < n + 1; ++i) cdf[i] /= funcInt;
    }
    int Count() const { return (int)func.size(); }
    int SampleDiscrete(Float u, Float *pdf) const {
        int size = (int)cdf.size();
        int first = 0, len = size;    // FindInterval, pbrt.h:403-415
        while (len > 0) {
            int half = len >> 1, middle = first + half;
            if (cdf[middle] <= u) { first = middle + 1; len -= half + 1; }
            else len = half;
        }
        int offset = Clamp(first - 1, 0, size - 2);
        if (pdf) *pdf = (funcInt > 0) ? func[offset] / (funcInt * Count()) : 0;
        return offset;
    }
};

// Distribution1D::SampleContinuous (core/sampling.h:85-103) and Distribution2D (core/sampling.h:128-151, core/sampling.cpp:97-108)
inline Float SampleContinuous1D(const Distribution1D &d, Float u, Float *pdf, int *off = nullptr) {
    int size = (int)d.cdf.size();
    int first = 0, len = size;    // FindInterval, pbrt.h:403-415
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (d.cdf[middle] <= u) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    int offset = Clamp(first - 1, 0, size - 2);
    if (off) *off = offset;
    Float du = u - d.cdf[offset];
    if ((d.cdf[offset + 1] - d.cdf[offset]) > 0) du /= (d.cdf[offset + 1] - d.cdf[offset]);
    if (pdf) *pdf = (d.funcInt > 0) ? d.func[offset] / d.funcInt : 0;
    return (offset + du) / d.Count();
}
struct Distribution2D {
    std::vector<Distribution1D> conditional;
    Distribution1D marginal;
    void Init(const Float *func, int nu, int nv) {
        conditional.resize(nv);
        for (int v = 0; v < nv; ++v) conditional[v].Init(&func[(size_t)v * nu], nu);
        std::vector<Float> m(nv);
        for (int v = 0; v < nv; ++v) m[v] = conditional[v].funcInt;
        marginal.Init(m.data(), nv);
    }
    P2 SampleContinuous(const P2 &u, Float *pdf) const {
        Float pdfs[2]; int v;
        Float d1 = SampleContinuous1D(marginal, u.y, &pdfs[1], &v);
        Float d0 = SampleContinuous1D(conditional[v], u.x, &pdfs[0]);
        *pdf = pdfs[0] * pdfs[1];
        P2 r; r.x = d0; r.y = d1; return r;
    }
    Float Pdf(const P2 &p) const {
        int iu = Clamp(int(p.x * conditional[0].Count()), 0, conditional[0].Count() - 1);
        int iv = Clamp(int(p.y * marginal.Count()), 0, marginal.Count() - 1);
        return conditional[iv].func[iu] / marginal.funcInt;
    }
};

struct Renderer {
    Scene scene;
    BVH bvh;                       // top-level aggregate
    std::vector<BVH> objectBvh;    // one per object definition (core/api.cpp:1798-1806)
    Camera camera;
    Film film;
    Distribution1D lightDistrib;          // "uniform" (or a single light) and "power": one distribution for every point
    // "spatial" (the reference's default with more than one light): SpatialLightDistribution, core/lightdistrib.cpp:77-300 — a
    // distribution per voxel of a grid over the world bound, computed when a vertex first falls into the voxel.  (The
    // reference's lock-free hash table is only its cache: a voxel's distribution is a pure function of the voxel.)
    bool spatial = false;
    int nVoxels[3] = {1, 1, 1};
    B3 worldBound;
    mutable std::mutex voxelMutex;
    mutable std::unordered_map<uint64_t, std::unique_ptr<Distribution1D>> voxelDist;
    Float worldRadius = 0; V3 worldCenter;
    std::vector<Distribution2D> envDist;      // per light; filled for infinite lights (lights/infinite.cpp:66-85)
    bool hasInfinite = false;
    int nThreads = 1;
    Counters total;
    double renderSeconds = 0;

    bool Setup(std::string *err) {
        if (scene.prm.filterType != 0) { *err = "only the box filter is in scope"; return false; }
        for (size_t li = 0; li < scene.lights.size(); ++li) {
            const Light &l = scene.lights[li];
            if (l.type != LIGHT_AREA) continue;
            const ShapeRec &sh = scene.shapes[l.shape];
            // a sphere's one light, or the per-triangle lights of an emissive mesh (consecutive, in face order)
            const bool own = sh.kind == SHAPE_SPHERE ? sh.areaLight == (int)li
                                                     : sh.areaLight >= 0 && (int)li >= sh.areaLight && (uint32_t)((int)li - sh.areaLight) < sh.nPrims;
            if (!own) { *err = "area light and its shape do not reference each other"; return false; }
        }
        objectBvh.resize(scene.objectPrims.size());
        uint32_t base = (uint32_t)scene.prims.size();
        for (size_t o = 0; o < objectBvh.size(); ++o) {
            objectBvh[o].Build(&scene, &scene.objectPrims[o], &objectBvh, base);
            base += (uint32_t)scene.objectPrims[o].size();
        }
        bvh.Build(&scene, &scene.prims, &objectBvh, 0);
        camera.Init(scene.prm);
        film.Init(scene.prm);
        // Scene ctor + Light::Preprocess (core/scene.h:56-66, lights/distant.h:57-59)
        B3 wb = bvh.WorldBound();
        worldBound = wb;
        worldCenter = (wb.pMin + wb.pMax) / 2;
        bool inside = worldCenter.x >= wb.pMin.x && worldCenter.x <= wb.pMax.x && worldCenter.y >= wb.pMin.y &&
                      worldCenter.y <= wb.pMax.y && worldCenter.z >= wb.pMin.z && worldCenter.z <= wb.pMax.z;
        worldRadius = inside ? Distance(worldCenter, wb.pMax) : 0;
        // InfiniteAreaLight's sampling distribution: a (2w x 2h) image of the map's luminance times sin(theta) (lights/infinite.cpp:66-85)
        envDist.resize(scene.lights.size());
        for (size_t li = 0; li < scene.lights.size(); ++li) {
            const Light &l = scene.lights[li];
            if (l.type != LIGHT_INFINITE) continue;
            if (l.tex < 0 || l.tex >= (int)scene.textures.size()) { *err = "infinite light without a map"; return false; }
            hasInfinite = true;
            const Texture &tx = scene.textures[l.tex];
            const int width = 2 * tx.levels[0].w, height = 2 * tx.levels[0].h;
            std::vector<Float> img((size_t)width * height);
            const float fwidth = 0.5f / smin(width, height);
            for (int v = 0; v < height; ++v) {
                Float vp = (v + .5f) / (Float)height;
                Float sinTheta = m_sinf(Pi * (v + .5f) / height);
                for (int u = 0; u < width; ++u) {
                    Float up = (u + .5f) / (Float)width;
                    P2 st; st.x = up; st.y = vp;
                    img[u + (size_t)v * width] = MipLookupWidth(tx, st, fwidth).y();
                    img[u + (size_t)v * width] *= sinTheta;
                }
            }
            envDist[li].Init(img.data(), width, height);
        }
        // CreateLightSampleDistribution, core/lightdistrib.cpp:48-66
        const int strategy = scene.lights.size() <= 1 ? 0 : scene.prm.lightStrategy;
        std::vector<Float> prob(smax<size_t>(1, scene.lights.size()), Float(1));
        if (strategy == 1)      // ComputeLightPowerDistribution, core/integrator.cpp:219-227
            for (size_t i = 0; i < scene.lights.size(); ++i) prob[i] = LightPower(scene.lights[i]).y();
        lightDistrib.Init(prob.data(), (int)scene.lights.size());
        spatial = strategy == 2;
        if (spatial) {          // SpatialLightDistribution::SpatialLightDistribution, core/lightdistrib.cpp:95-120 (maxVoxels = 64)
            V3 diag = wb.pMax - wb.pMin;
            int me = (diag.x > diag.y && diag.x > diag.z) ? 0 : (diag.y > diag.z ? 1 : 2);      // Bounds3::MaximumExtent, geometry.h:942-950
            Float bmax = diag[me];
            for (int i = 0; i < 3; ++i) nVoxels[i] = smax(1, int(std::round(diag[i] / bmax * 64)));
        }
        return true;
    }

    bool SceneIntersect(const Ray &ray, SurfaceInteraction *isect, Counters &ctr) const {   // core/scene.cpp:45-49
        ++ctr.rays;
        return bvh.Intersect(ray, isect, ctr);
    }
    bool SceneIntersectP(const Ray &ray, Counters &ctr) const {                             // core/scene.cpp:51-55
        ++ctr.shadowRays;
        return bvh.IntersectP(ray, ctr);
    }
    // DiffuseAreaLight::L, lights/diffuse.h:56-58
    static Spec AreaL(const Light &l, const V3 &n, const V3 &w) {
        return (l.twoSided || Dot(n, w) > 0) ? l.I : Spec(0.f);
    }
    // SurfaceInteraction::Le, core/interaction.cpp:151-154
    // GeometricPrimitive::GetAreaLight of the hit primitive: a sphere's light, or the light of the hit TRIANGLE of an emissive
    // mesh (every triangle is a Shape with a DiffuseAreaLight of its own, core/api.cpp:1609-1636; a mesh's lights are consecutive)
    int AreaLightOf(const SurfaceInteraction &isect) const {
        const ShapeRec &sh = scene.shapes[isect.shape];
        if (sh.areaLight < 0) return -1;
        return sh.kind == SHAPE_MESH ? sh.areaLight + isect.tri : sh.areaLight;
    }
    Spec Le(const SurfaceInteraction &isect, const V3 &w) const {
        int al = AreaLightOf(isect);
        return al >= 0 ? AreaL(scene.lights[al], isect.n, w) : Spec(0.f);
    }
    // SphericalTheta / SphericalPhi, core/geometry.h:1816-1824
    static Float SphericalTheta(const V3 &v) { return m_acosf(Clamp(v.z, -1, 1)); }
    static Float SphericalPhi(const V3 &v) { Float p = m_atan2f(v.y, v.x); return (p < 0) ? (p + 2 * Pi) : p; }
    // Light::Le(ray): radiance an escaped ray picks up (core/light.cpp:66; InfiniteAreaLight::Le, lights/infinite.cpp:93-97)
    Spec LightLe(const Light &l, const V3 &d) const {
        if (l.type != LIGHT_INFINITE) return Spec(0.f);
        V3 w = Normalize(XfVector(l.w2l, d));
        P2 st; st.x = SphericalPhi(w) * Inv2Pi; st.y = SphericalTheta(w) * InvPi;
        return MipLookupWidth(scene.textures[l.tex], st, 0.f);
    }
    // Light::Power: lights/point.cpp:55, lights/distant.cpp:61-63, lights/diffuse.cpp:64-66 (area = shape->Area())
    Spec LightPower(const Light &l) const {
        if (l.type == LIGHT_POINT) return 4 * Pi * l.I;
        if (l.type == LIGHT_DISTANT) return l.I * Pi * worldRadius * worldRadius;
        if (l.type == LIGHT_INFINITE) { P2 c; c.x = .5f; c.y = .5f; return Pi * worldRadius * worldRadius * MipLookupWidth(scene.textures[l.tex], c, .5f); }   // lights/infinite.cpp:87-91
        const ShapeRec &sh = scene.shapes[l.shape];
        Float area;
        if (sh.kind == SHAPE_MESH) {
            const Mesh &m = scene.meshes[sh.meshIndex];
            const int *v = &m.idx[3 * ((int)(&l - scene.lights.data()) - sh.areaLight)];
            area = TriangleArea(m.p[v[0]], m.p[v[1]], m.p[v[2]]);
        } else { const Sphere &s = scene.spheres[sh.sphereIndex]; area = s.phiMax * s.radius * (s.zMax - s.zMin); }
        return (Float)(l.twoSided ? 2 : 1) * l.I * area * Pi;
    }
    // LightDistribution::Lookup(p)
    const Distribution1D &LookupLightDistribution(const V3 &p) const {
        if (!spatial) return lightDistrib;
        // SpatialLightDistribution::Lookup, core/lightdistrib.cpp:134-147: Bounds3::Offset, then int(offset * nVoxels) clamped
        V3 o = p - worldBound.pMin;
        if (worldBound.pMax.x > worldBound.pMin.x) o.x /= worldBound.pMax.x - worldBound.pMin.x;
        if (worldBound.pMax.y > worldBound.pMin.y) o.y /= worldBound.pMax.y - worldBound.pMin.y;
        if (worldBound.pMax.z > worldBound.pMin.z) o.z /= worldBound.pMax.z - worldBound.pMin.z;
        int pi[3];
        for (int i = 0; i < 3; ++i) pi[i] = Clamp(int(o[i] * nVoxels[i]), 0, nVoxels[i] - 1);
        const uint64_t packed = ((uint64_t)pi[0] << 40) | ((uint64_t)pi[1] << 20) | (uint64_t)pi[2];
        std::lock_guard<std::mutex> lock(voxelMutex);
        auto it = voxelDist.find(packed);
        if (it != voxelDist.end()) return *it->second;
        // ComputeDistribution, core/lightdistrib.cpp:231-298
        V3 p0(Float(pi[0]) / Float(nVoxels[0]), Float(pi[1]) / Float(nVoxels[1]), Float(pi[2]) / Float(nVoxels[2]));
        V3 p1(Float(pi[0] + 1) / Float(nVoxels[0]), Float(pi[1] + 1) / Float(nVoxels[1]), Float(pi[2] + 1) / Float(nVoxels[2]));
        auto wlerp = [&](const V3 &t) { return V3(Lerp(t.x, worldBound.pMin.x, worldBound.pMax.x), Lerp(t.y, worldBound.pMin.y, worldBound.pMax.y),
                                                  Lerp(t.z, worldBound.pMin.z, worldBound.pMax.z)); };
        const V3 vMin = wlerp(p0), vMax = wlerp(p1);
        const int nSamples = 128;
        std::vector<Float> lightContrib(scene.lights.size(), Float(0));
        for (int i = 0; i < nSamples; ++i) {
            V3 t(RadicalInverse(0, i), RadicalInverse(1, i), RadicalInverse(2, i));
            Interaction intr;
            intr.p = V3(Lerp(t.x, vMin.x, vMax.x), Lerp(t.y, vMin.y, vMax.y), Lerp(t.z, vMin.z, vMax.z));
            intr.pError = V3(); intr.n = V3();
            P2 u(RadicalInverse(3, i), RadicalInverse(4, i));
            for (size_t j = 0; j < scene.lights.size(); ++j) {
                Float pdf; V3 wi; Interaction pl;
                Spec Li = Sample_Li(scene.lights[j], intr, u, &wi, &pdf, &pl);
                if (pdf > 0) lightContrib[j] += Li.y() / pdf;
            }
        }
        Float sumContrib = 0;
        for (Float c : lightContrib) sumContrib += c;      // std::accumulate(..., Float(0))
        Float avgContrib = sumContrib / (nSamples * lightContrib.size());
        Float minContrib = (avgContrib > 0) ? (Float)(.001 * (double)avgContrib) : 1;
        for (Float &c : lightContrib) c = smax(c, minContrib);
        std::unique_ptr<Distribution1D> d(new Distribution1D());
        d->Init(lightContrib.data(), (int)lightContrib.size());
        return *(voxelDist[packed] = std::move(d));
    }
    // Light::Sample_Li for the three light types
    Spec Sample_Li(const Light &l, const Interaction &ref, const P2 &u, V3 *wi, Float *pdf, Interaction *pLight) const {
        if (l.type == LIGHT_POINT) {          // lights/point.cpp:44-53
            *wi = Normalize(l.pos - ref.p);
            *pdf = 1.f;
            pLight->p = l.pos; pLight->pError = V3(); pLight->n = V3();
            return l.I / DistanceSquared(l.pos, ref.p);
        } else if (l.type == LIGHT_DISTANT) { // lights/distant.cpp:49-59
            *wi = l.pos;
            *pdf = 1;
            V3 pOutside = ref.p + l.pos * (2 * worldRadius);
            pLight->p = pOutside; pLight->pError = V3(); pLight->n = V3();
            return l.I;
        } else if (l.type == LIGHT_INFINITE) { // lights/infinite.cpp:99-124
            Float mapPdf;
            P2 uv = envDist[&l - scene.lights.data()].SampleContinuous(u, &mapPdf);
            if (mapPdf == 0) { *pdf = 0; return Spec(0.f); }      // (the reference leaves *pdf untouched: its caller initialised it to 0)
            Float theta = uv.y * Pi, phi = uv.x * 2 * Pi;
            Float cosTheta = m_cosf(theta), sinTheta = m_sinf(theta);
            Float sinPhi = m_sinf(phi), cosPhi = m_cosf(phi);
            *wi = XfVector(l.l2w, V3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta));
            *pdf = mapPdf / (2 * Pi * Pi * sinTheta);
            if (sinTheta == 0) *pdf = 0;
            pLight->p = ref.p + *wi * (2 * worldRadius); pLight->pError = V3(); pLight->n = V3();
            return MipLookupWidth(scene.textures[l.tex], uv, 0.f);
        } else {                              // lights/diffuse.cpp:68-81
            const ShapeRec &sh = scene.shapes[l.shape];
            Interaction pShape;
            if (sh.kind == SHAPE_MESH) {
                const Mesh &m = scene.meshes[sh.meshIndex];
                const int tri = (int)(&l - scene.lights.data()) - sh.areaLight;
                pShape = TriangleSample(m, &m.idx[3 * tri], (sh.reverseOrientation != 0) ^ (sh.swapsHandedness != 0), ref, u, pdf);
            } else pShape = SphereSample(scene.spheres[sh.sphereIndex], sh.reverseOrientation != 0, ref, u, pdf);
            if (*pdf == 0 || (pShape.p - ref.p).LengthSquared() == 0) { *pdf = 0; return 0.f; }
            *wi = Normalize(pShape.p - ref.p);
            *pLight = pShape;
            return AreaL(l, pShape.n, -*wi);
        }
    }
    // Light::Pdf_Li: point/distant 0; area -> Sphere::Pdf (shapes/sphere.cpp:294-306)
    Float Pdf_Li(const Light &l, const Interaction &ref, const V3 &wi, Counters &ctr) const {
        if (l.type == LIGHT_INFINITE) {       // lights/infinite.cpp:126-134
            V3 w = XfVector(l.w2l, wi);
            Float theta = SphericalTheta(w), phi = SphericalPhi(w);
            Float sinTheta = m_sinf(theta);
            if (sinTheta == 0) return 0;
            P2 st; st.x = phi * Inv2Pi; st.y = theta * InvPi;
            return envDist[&l - scene.lights.data()].Pdf(st) / (2 * Pi * Pi * sinTheta);
        }
        if (l.type != LIGHT_AREA) return 0;
        const ShapeRec &sh = scene.shapes[l.shape];
        if (sh.kind == SHAPE_MESH) {
            // Shape::Pdf, core/shape.cpp:72-88, with Triangle::Intersect and Triangle::Area
            const Mesh &m = scene.meshes[sh.meshIndex];
            const int tri = (int)(&l - scene.lights.data()) - sh.areaLight;
            Ray ray = SpawnRay(ref.p, ref.pError, ref.n, wi);
            Float tHit; SurfaceInteraction isectLight;
            TriRef tr{&m, &m.idx[3 * tri], (bool)((sh.reverseOrientation != 0) ^ (sh.swapsHandedness != 0))};
            Counters own;      // (this test is not the aggregate's: orc_accel.h Counters)
            const bool found = TriangleIntersect(tr, ray, &tHit, &isectLight, own);
            ctr.triTestsPdf += own.triTests; ctr.triHitsPdf += own.triHits;
            if (!found) return 0;
            const int *v = &m.idx[3 * tri];
            Float pdf = DistanceSquared(ref.p, isectLight.p) / (AbsDot(isectLight.n, -wi) * TriangleArea(m.p[v[0]], m.p[v[1]], m.p[v[2]]));
            if (std::isinf(pdf)) pdf = 0.f;
            return pdf;
        }
        const Sphere &s = scene.spheres[sh.sphereIndex];
        V3 pCenter = XfPoint(s.o2w, V3(0, 0, 0));
        V3 pOrigin = OffsetRayOrigin(ref.p, ref.pError, ref.n, pCenter - ref.p);
        if (DistanceSquared(pOrigin, pCenter) <= s.radius * s.radius) {
            // Shape::Pdf, core/shape.cpp:72-88
            Ray ray = SpawnRay(ref.p, ref.pError, ref.n, wi);
            Float tHit; SurfaceInteraction isectLight;
            bool flip = (sh.reverseOrientation != 0) ^ (sh.swapsHandedness != 0);
            Counters own;
            const bool found = SphereIntersect(s, flip, ray, &tHit, &isectLight, own);
            ctr.sphereTestsPdf += own.sphereTests;
            if (!found) return 0;
            Float area = s.phiMax * s.radius * (s.zMax - s.zMin);
            Float pdf = DistanceSquared(ref.p, isectLight.p) / (AbsDot(isectLight.n, -wi) * area);
            if (std::isinf(pdf)) pdf = 0.f;
            return pdf;
        }
        Float sinThetaMax2 = s.radius * s.radius / DistanceSquared(ref.p, pCenter);
        Float cosThetaMax = std::sqrt(smax((Float)0, 1 - sinThetaMax2));
        return UniformConePdf(cosThetaMax);
    }

    // core/integrator.cpp:109-217 (surface interactions, handleMedia=false, specular=false)
    Spec EstimateDirect(const SurfaceInteraction &isect, const BSDF &bsdf, const P2 &uScattering, int lightNum,
                        const P2 &uLight, Counters &ctr) const {
        const Light &light = scene.lights[lightNum];
        const int bsdfFlags = BSDF_ALL & ~BSDF_SPECULAR;
        Spec Ld(0.f);
        V3 wi;
        Float lightPdf = 0, scatteringPdf = 0;
        Interaction it{isect.p, isect.pError, isect.n};
        Interaction pLight;
        Spec Li = Sample_Li(light, it, uLight, &wi, &lightPdf, &pLight);
        bool isDelta = (light.type == LIGHT_POINT || light.type == LIGHT_DISTANT);      // IsDeltaLight(flags): an infinite light is not
        if (lightPdf > 0 && !Li.IsBlack()) {
            Spec f = bsdf.f(isect.wo, wi, bsdfFlags) * AbsDot(wi, isect.shading.n);
            scatteringPdf = bsdf.Pdf(isect.wo, wi, bsdfFlags);
            if (!f.IsBlack()) {
                Ray r = SpawnRayTo(it, pLight);     // VisibilityTester::Unoccluded, core/light.cpp:59-64
                if (SceneIntersectP(r, ctr)) Li = Spec(0.f);
                if (!Li.IsBlack()) {
                    if (isDelta) Ld += f * Li / lightPdf;
                    else {
                        Float weight = PowerHeuristic(1, lightPdf, 1, scatteringPdf);
                        Ld += f * Li * weight / lightPdf;
                    }
                }
            }
        }
        if (!isDelta) {
            Spec f;
            int sampledType = 0;
            f = bsdf.Sample_f(isect.wo, &wi, uScattering, &scatteringPdf, bsdfFlags, &sampledType);
            f *= AbsDot(wi, isect.shading.n);
            bool sampledSpecular = (sampledType & BSDF_SPECULAR) != 0;
            if (!f.IsBlack() && scatteringPdf > 0) {
                Float weight = 1;
                if (!sampledSpecular) {
                    lightPdf = Pdf_Li(light, it, wi, ctr);
                    if (lightPdf == 0) return Ld;
                    weight = PowerHeuristic(1, scatteringPdf, 1, lightPdf);
                }
                SurfaceInteraction lightIsect;
                Ray ray = SpawnRay(isect.p, isect.pError, isect.n, wi);
                bool found = SceneIntersect(ray, &lightIsect, ctr);
                Spec Li2(0.f);
                if (found) {
                    if (AreaLightOf(lightIsect) == lightNum) Li2 = Le(lightIsect, -wi);
                } else Li2 = LightLe(light, ray.d);      // light.Le(ray): zero except for infinite lights (core/light.cpp:66, lights/infinite.cpp:93-97)
                if (!Li2.IsBlack()) Ld += f * Li2 * Spec(1.f) * weight / scatteringPdf;
            }
        }
        return Ld;
    }
    // core/integrator.cpp:86-107
    Spec UniformSampleOneLight(const SurfaceInteraction &isect, const BSDF &bsdf, HaltonSampler &sampler,
                               Counters &ctr) const {
        int nLights = (int)scene.lights.size();
        if (nLights == 0) return Spec(0.f);
        Float lightPdf;
        int lightNum = LookupLightDistribution(isect.p).SampleDiscrete(sampler.Get1D(), &lightPdf);      // path.cpp:125
        if (lightPdf == 0) return Spec(0.f);
        P2 uLight = sampler.Get2D();
        P2 uScattering = sampler.Get2D();
        return EstimateDirect(isect, bsdf, uScattering, lightNum, uLight, ctr) / lightPdf;
    }
    // integrators/path.cpp:64-204 (no media, no BSSRDF; every material has a BSDF)
    Spec Li(const Ray &r, HaltonSampler &sampler, Counters &ctr, int *pathLen = nullptr, const RayDiff *camDiff = nullptr) const {
        RayDiff rdiff; if (camDiff) rdiff = *camDiff;     // only the camera ray has differentials: SpawnRay returns a plain Ray
        Spec L(0.f), beta(1.f);
        Ray ray(r);
        bool specularBounce = false;
        int bounces;
        Float etaScale = 1;
        const int maxDepth = scene.prm.maxDepth;
        const Float rrThreshold = scene.prm.rrThreshold;
        for (bounces = 0;; ++bounces) {
            SurfaceInteraction isect;
            bool foundIntersection = SceneIntersect(ray, &isect, ctr);
            if (bounces == 0 || specularBounce) {
                if (foundIntersection) L += beta * Le(isect, -ray.d);
                else for (const Light &l : scene.lights) if (l.type == LIGHT_INFINITE) L += beta * LightLe(l, ray.d);      // scene.infiniteLights, path.cpp:104-106
            }
            if (!foundIntersection || bounces >= maxDepth) break;
            BSDF bsdf;
            ComputeDifferentials(&isect, ray, rdiff);      // SurfaceInteraction::ComputeScatteringFunctions, core/interaction.cpp:93-101
            ComputeScatteringFunctions(scene, scene.materials[scene.shapes[isect.shape].material], isect, &bsdf);
            if (bsdf.NumComponents(BSDF_ALL & ~BSDF_SPECULAR) > 0) {
                Spec Ld = beta * UniformSampleOneLight(isect, bsdf, sampler, ctr);
                L += Ld;
            }
            V3 wo = -ray.d, wi;
            Float pdf = 0;
            int flags = 0;
            Spec f = bsdf.Sample_f(wo, &wi, sampler.Get2D(), &pdf, BSDF_ALL, &flags);
            if (f.IsBlack() || pdf == 0.f) break;
            beta *= f * AbsDot(wi, isect.shading.n) / pdf;
            specularBounce = (flags & BSDF_SPECULAR) != 0;
            if ((flags & BSDF_SPECULAR) && (flags & BSDF_TRANSMISSION)) {      // path.cpp:154-162
                Float eta = bsdf.eta;
                etaScale *= (Dot(wo, isect.n) > 0) ? (eta * eta) : 1 / (eta * eta);
            }
            ray = SpawnRay(isect.p, isect.pError, isect.n, wi);
            rdiff.has = false;
            Spec rrBeta = beta * etaScale;
            if (rrBeta.MaxComponentValue() < rrThreshold && bounces > 3) {
                Float q = smax((Float).05, 1 - rrBeta.MaxComponentValue());
                if (sampler.Get1D() < q) break;
                beta /= 1 - q;
            }
        }
        if (pathLen) *pathLen = bounces;
        return L;
    }

    // One camera sample: core/integrator.cpp:281-333 (minus the film add)
    Spec RenderSample(HaltonSampler &sampler, int px, int py, P2 *pFilmOut, Float *rayWeightOut, Counters &ctr) const {
        P2 u = sampler.Get2D();                       // Sampler::GetCameraSample, core/sampler.cpp:46-52
        P2 pFilm((Float)px + u.x, (Float)py + u.y);
        Float time = sampler.Get1D(); (void)time;
        P2 pLens = sampler.Get2D();
        Ray ray; RayDiff rd;
        const bool textured = !scene.textures.empty();
        Float rayWeight = camera.GenerateRay(pFilm, pLens, &ray, textured ? &rd : nullptr);
        if (textured) rd.Scale(ray, 1 / std::sqrt((Float)sampler.samplesPerPixel));     // core/integrator.cpp:288-289
        ++ctr.cameraRays;
        Spec L(0.f);
        if (rayWeight > 0) L = Li(ray, sampler, ctr, nullptr, textured ? &rd : nullptr);
        if (L.HasNaNs()) L = Spec(0.f);
        else if (L.y() < -1e-5) L = Spec(0.f);
        else if (std::isinf(L.y())) L = Spec(0.f);
        *pFilmOut = pFilm; *rayWeightOut = rayWeight;
        return L;
    }

    // SamplerIntegrator::Render, core/integrator.cpp:230-360.  spp<=0 uses the scene's.
    void Render(int sppOverride = 0, int tileBegin = 0, int tileStride = 1) {
        int spp = sppOverride > 0 ? sppOverride : scene.prm.spp;
        int sx0, sy0, sx1, sy1;
        film.GetSampleBounds(&sx0, &sy0, &sx1, &sy1);
        for (auto &p : film.pixels) p = FilmPixel();
        for (auto &p : film.stats) p = std::array<uint64_t, 7>{{0, 0, 0, 0, 0, 0, 0}};
        const int tileSize = 16;
        int ntx = (sx1 - sx0 + tileSize - 1) / tileSize, nty = (sy1 - sy0 + tileSize - 1) / tileSize;
        std::atomic<int> next(0);
        total = Counters();
        std::mutex cmutex;
        auto t0 = std::chrono::high_resolution_clock::now();
        auto worker = [&]() {
            Counters ctr;
            HaltonSampler sampler(spp, sx0, sy0, sx1, sy1, scene.prm.samplePixelCenter != 0);
            while (true) {
                int t = tileBegin + tileStride * next.fetch_add(1);   // tile subset: what one GPU of a tile-sharded job renders
                if (t >= ntx * nty) break;
                int tx = t % ntx, ty = t / ntx;
                int x0 = sx0 + tx * tileSize, x1 = smin(x0 + tileSize, sx1);
                int y0 = sy0 + ty * tileSize, y1 = smin(y0 + tileSize, sy1);
                FilmTile tile(&film, x0, y0, x1, y1);
                for (int y = y0; y < y1; ++y)
                    for (int x = x0; x < x1; ++x) {
                        sampler.StartPixel(x, y);
                        const Counters before = ctr;
                        // pixelBounds == sample bounds for PathIntegrator without "pixelbounds" (path.cpp:212)
                        do {
                            P2 pFilm; Float w;
                            Spec L = RenderSample(sampler, x, y, &pFilm, &w, ctr);
                            tile.AddSample(pFilm, L, w);
                        } while (sampler.StartNextSample());
                        // filmTile->GetPixel(pixel).stats += ray.stats (core/integrator.cpp:327-328): every ray of the
                        // pixel's samples adds its counters once (integrators/path.cpp:92-200, core/integrator.cpp:213,
                        // core/light.cpp:62); MergeFilmTile sums them into the film (core/film.cpp:130)
                        std::array<uint64_t, 7> &ps = film.stats[(size_t)(x - film.cx0) + (size_t)(y - film.cy0) * (film.cx1 - film.cx0)];
                        ps[0] += ctr.cameraRays - before.cameraRays;
                        ps[1] += (ctr.triTests + ctr.sphereTests) - (before.triTests + before.sphereTests);
                        ps[2] += (ctr.triTestsP + ctr.sphereTestsP) - (before.triTestsP + before.sphereTestsP);
                        ps[3] += ctr.leavesEntered - before.leavesEntered;
                        ps[4] += ctr.leavesEnteredP - before.leavesEnteredP;
                        ps[5] += (ctr.nodesEntered - ctr.leavesEntered) - (before.nodesEntered - before.leavesEntered);
                        ps[6] += (ctr.nodesEnteredP - ctr.leavesEnteredP) - (before.nodesEnteredP - before.leavesEnteredP);
                    }
                MergeFilmTile(&film, tile);
            }
            std::lock_guard<std::mutex> lk(cmutex);
            total.add(ctr);
        };
        std::vector<std::thread> th;
        for (int i = 1; i < nThreads; ++i) th.emplace_back(worker);
        worker();
        for (auto &t : th) t.join();
        renderSeconds = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    }
};

}  // namespace orc
