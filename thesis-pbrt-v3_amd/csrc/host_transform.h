// hprt host side — static 4x4 transforms with explicit inverse, arithmetic as in
// the pbrt-v3 fork's core/transform.cpp (cited per function).  sin/cos/tan here are
// the host libm's, as in the reference: they run once per scene, never per ray.
#pragma once
#include <cmath>
#include <utility>
#include "hprt_math.h"

namespace hprt {

inline mat4 mat_identity() {
    mat4 r;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = (i == j) ? 1.f : 0.f;
    return r;
}
inline mat4 mat_mul(const mat4 &a, const mat4 &b) {   // Matrix4x4::Mul, transform.h:83-91
    mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
    return r;
}
inline mat4 mat_transpose(const mat4 &a) {
    mat4 r;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = a.m[j][i];
    return r;
}
// Gauss-Jordan elimination with full pivoting, transform.cpp:82-139.  The pivot
// reciprocal is formed in double ("1. / x") and rounded to float, as there.
inline bool mat_inverse(const mat4 &src, mat4 *out) {
    int colIdx[4], rowIdx[4], pivoted[4] = {0, 0, 0, 0};
    float a[4][4];
    memcpy(a, src.m, sizeof(a));
    for (int step = 0; step < 4; ++step) {
        int prow = 0, pcol = 0;
        float best = 0.f;
        for (int j = 0; j < 4; ++j) {
            if (pivoted[j] == 1) continue;
            for (int k = 0; k < 4; ++k) {
                if (pivoted[k] == 0) {
                    if (std::fabs(a[j][k]) >= best) { best = std::fabs(a[j][k]); prow = j; pcol = k; }
                } else if (pivoted[k] > 1) return false;
            }
        }
        ++pivoted[pcol];
        if (prow != pcol) for (int k = 0; k < 4; ++k) std::swap(a[prow][k], a[pcol][k]);
        rowIdx[step] = prow; colIdx[step] = pcol;
        if (a[pcol][pcol] == 0.f) return false;
        float pinv = (float)(1. / a[pcol][pcol]);
        a[pcol][pcol] = 1.f;
        for (int j = 0; j < 4; ++j) a[pcol][j] *= pinv;
        for (int j = 0; j < 4; ++j) {
            if (j == pcol) continue;
            float save = a[j][pcol];
            a[j][pcol] = 0;
            for (int k = 0; k < 4; ++k) a[j][k] -= a[pcol][k] * save;
        }
    }
    for (int j = 3; j >= 0; --j)
        if (rowIdx[j] != colIdx[j]) for (int k = 0; k < 4; ++k) std::swap(a[k][rowIdx[j]], a[k][colIdx[j]]);
    memcpy(out->m, a, sizeof(a));
    return true;
}

struct Xform {
    mat4 m, inv;
    Xform() : m(mat_identity()), inv(mat_identity()) {}
    Xform(const mat4 &m_, const mat4 &inv_) : m(m_), inv(inv_) {}
    explicit Xform(const mat4 &m_) : m(m_) { if (!mat_inverse(m_, &inv)) inv = mat_identity(); }
    Xform operator*(const Xform &o) const { return Xform(mat_mul(m, o.m), mat_mul(o.inv, inv)); }   // transform.cpp:251-253
    Xform inverse() const { return Xform(inv, m); }
    bool swapsHandedness() const {      // transform.cpp:255-260
        float det = m.m[0][0] * (m.m[1][1] * m.m[2][2] - m.m[1][2] * m.m[2][1]) -
                    m.m[0][1] * (m.m[1][0] * m.m[2][2] - m.m[1][2] * m.m[2][0]) +
                    m.m[0][2] * (m.m[1][0] * m.m[2][1] - m.m[1][1] * m.m[2][0]);
        return det < 0;
    }
};
inline float radians(float deg) { return (HPRT_PI / 180) * deg; }   // pbrt.h:324
inline Xform xf_translate(vec3 d) {                                  // transform.cpp:141-147
    Xform t;
    t.m.m[0][3] = d.x; t.m.m[1][3] = d.y; t.m.m[2][3] = d.z;
    t.inv.m[0][3] = -d.x; t.inv.m[1][3] = -d.y; t.inv.m[2][3] = -d.z;
    return t;
}
inline Xform xf_scale(float x, float y, float z) {                   // transform.cpp:149-153
    Xform t;
    t.m.m[0][0] = x; t.m.m[1][1] = y; t.m.m[2][2] = z;
    t.inv.m[0][0] = 1 / x; t.inv.m[1][1] = 1 / y; t.inv.m[2][2] = 1 / z;
    return t;
}
inline Xform xf_rotate(float theta, vec3 axis) {                     // transform.cpp:179-201
    vec3 a = normalize(axis);
    float s = std::sin(radians(theta)), c = std::cos(radians(theta));
    mat4 m = mat_identity();
    m.m[0][0] = a.x * a.x + (1 - a.x * a.x) * c;
    m.m[0][1] = a.x * a.y * (1 - c) - a.z * s;
    m.m[0][2] = a.x * a.z * (1 - c) + a.y * s;
    m.m[0][3] = 0;
    m.m[1][0] = a.x * a.y * (1 - c) + a.z * s;
    m.m[1][1] = a.y * a.y + (1 - a.y * a.y) * c;
    m.m[1][2] = a.y * a.z * (1 - c) - a.x * s;
    m.m[1][3] = 0;
    m.m[2][0] = a.x * a.z * (1 - c) - a.y * s;
    m.m[2][1] = a.y * a.z * (1 - c) + a.x * s;
    m.m[2][2] = a.z * a.z + (1 - a.z * a.z) * c;
    m.m[2][3] = 0;
    return Xform(m, mat_transpose(m));
}
inline bool xf_look_at(vec3 pos, vec3 look, vec3 up, Xform *out) {   // transform.cpp:203-235
    mat4 c2w = mat_identity();
    c2w.m[0][3] = pos.x; c2w.m[1][3] = pos.y; c2w.m[2][3] = pos.z; c2w.m[3][3] = 1;
    vec3 dir = normalize(look - pos);
    if (length(cross(normalize(up), dir)) == 0) { *out = Xform(); return false; }
    vec3 right = normalize(cross(normalize(up), dir));
    vec3 newUp = cross(dir, right);
    c2w.m[0][0] = right.x; c2w.m[1][0] = right.y; c2w.m[2][0] = right.z; c2w.m[3][0] = 0.;
    c2w.m[0][1] = newUp.x; c2w.m[1][1] = newUp.y; c2w.m[2][1] = newUp.z; c2w.m[3][1] = 0.;
    c2w.m[0][2] = dir.x; c2w.m[1][2] = dir.y; c2w.m[2][2] = dir.z; c2w.m[3][2] = 0.;
    mat4 w2c;
    if (!mat_inverse(c2w, &w2c)) w2c = mat_identity();
    *out = Xform(w2c, c2w);
    return true;
}
inline Xform xf_perspective(float fov, float n, float f) {           // transform.cpp:303-311
    mat4 persp = mat_identity();
    persp.m[2][2] = f / (f - n); persp.m[2][3] = -f * n / (f - n);
    persp.m[3][2] = 1; persp.m[3][3] = 0;
    float invTan = 1 / std::tan(radians(fov) / 2);
    return xf_scale(invTan, invTan, 1) * Xform(persp);
}

}  // namespace hprt
