// hprt — scalar/vector arithmetic shared by the host front-end and the HIP kernels.
//
// Every function here reproduces, operation for operation, the IEEE-754 single
// rounding sequence of the pbrt-v3 fork's core/geometry.h / core/pbrt.h helpers
// it names, because ray/triangle hit selection and the radiance estimate must
// agree with the CPU PathIntegrator bit for bit.  Translation units including
// this header are built with -ffp-contract=off and without fast-math; the
// device keeps f32 denormals and uses correctly rounded f32 divide/sqrt
// (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HPRT_HD __host__ __device__ __forceinline__
#else
#define HPRT_HD inline
#endif

namespace hprt {

// constants as *float* values (reference: core/pbrt.h:193-210, core/rng.h:52)
#define HPRT_INF (__builtin_huge_valf())
#define HPRT_MACHINE_EPS 5.9604644775390625e-08f /* 2^-24 */
#define HPRT_SHADOW_EPS 0.0001f
#define HPRT_PI 3.14159274101257324219f
#define HPRT_INV_PI 0.31830987334251403809f
#define HPRT_INV_2PI 0.15915494309189533577f
#define HPRT_PI_OVER_2 1.57079637050628662109f
#define HPRT_PI_OVER_4 0.78539818525314331055f
#define HPRT_ONE_MINUS_EPS 0.99999994039535522461f

HPRT_HD uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
HPRT_HD float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
HPRT_HD bool is_inf(float v) { return (f2u(v) & 0x7fffffffu) == 0x7f800000u; }
HPRT_HD bool is_nan(float v) { return (f2u(v) & 0x7fffffffu) > 0x7f800000u; }

// libstdc++ std::min/std::max select semantics (NaN-sensitive; never fminf)
HPRT_HD float sel_min(float a, float b) { return (b < a) ? b : a; }
HPRT_HD float sel_max(float a, float b) { return (a < b) ? b : a; }
HPRT_HD int sel_min(int a, int b) { return (b < a) ? b : a; }
HPRT_HD int sel_max(int a, int b) { return (a < b) ? b : a; }
HPRT_HD float clampf(float v, float lo, float hi) { return (v < lo) ? lo : ((v > hi) ? hi : v); }

// gamma(n) = n*eps/(1-n*eps), evaluated in float exactly as core/pbrt.h:289-291
HPRT_HD float gamma_n(int n) { return ((float)n * HPRT_MACHINE_EPS) / (1.0f - (float)n * HPRT_MACHINE_EPS); }

// core/pbrt.h:241-265
HPRT_HD float next_up(float v) {
    if (is_inf(v) && v > 0.f) return v;
    if (v == -0.f) v = 0.f;
    uint32_t ui = f2u(v);
    if (v >= 0) ++ui; else --ui;
    return u2f(ui);
}
HPRT_HD float next_down(float v) {
    if (is_inf(v) && v < 0.f) return v;
    if (v == 0.f) v = -0.f;
    uint32_t ui = f2u(v);
    if (v > 0) --ui; else ++ui;
    return u2f(ui);
}

// Distribution1D (core/sampling.h:55-109), the part the light pick of UniformSampleOneLight uses: the constructor's cdf
// (host) and SampleDiscrete with FindInterval (core/pbrt.h:403-415) and its pdf (host and device).
inline void dist1d_build(const float *func, int n, float *cdf, float *funcInt) {
    cdf[0] = 0;
    for (int i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / n;
    *funcInt = cdf[n];
    if (*funcInt == 0) for (int i = 1; i < n + 1; ++i) cdf[i] = float(i) / float(n);
    else for (int i = 1; i < n + 1; ++i) cdf[i] /= *funcInt;
}
HPRT_HD int dist1d_sample_discrete(const float *cdf, const float *func, float funcInt, int n, float u, float *pdf) {
    int size = n + 1;
    int first = 0, len = size;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (cdf[middle] <= u) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    int offset = first - 1;
    if (offset < 0) offset = 0; else if (offset > size - 2) offset = size - 2;
    *pdf = (funcInt > 0) ? func[offset] / (funcInt * n) : 0;
    return offset;
}

struct vec3 {
    float x, y, z;
    HPRT_HD vec3() : x(0.f), y(0.f), z(0.f) {}
    HPRT_HD vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    HPRT_HD float get(int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    HPRT_HD void set(int i, float v) { if (i == 0) x = v; else if (i == 1) y = v; else z = v; }
};
HPRT_HD vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
HPRT_HD vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
HPRT_HD vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
HPRT_HD vec3 operator*(float s, vec3 a) { return vec3(s * a.x, s * a.y, s * a.z); }
HPRT_HD vec3 operator*(vec3 a, float s) { return vec3(s * a.x, s * a.y, s * a.z); }
// geometry.h:281-286 — "v / f" multiplies by the float reciprocal
HPRT_HD vec3 div_by(vec3 a, float f) { float inv = 1.0f / f; return vec3(a.x * inv, a.y * inv, a.z * inv); }
HPRT_HD float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
HPRT_HD float absdot(vec3 a, vec3 b) { return fabsf(dot(a, b)); }
HPRT_HD float length2(vec3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
HPRT_HD float length(vec3 a) { return sqrtf(length2(a)); }
HPRT_HD vec3 normalize(vec3 a) { return div_by(a, length(a)); }
HPRT_HD vec3 vabs(vec3 a) { return vec3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
// geometry.h:1287-1321 — products and differences in double, one rounding to float
HPRT_HD vec3 cross(vec3 a, vec3 b) {
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return vec3((float)((ay * bz) - (az * by)), (float)((az * bx) - (ax * bz)), (float)((ax * by) - (ay * bx)));
}
HPRT_HD float dist2(vec3 a, vec3 b) { return length2(a - b); }
HPRT_HD float dist(vec3 a, vec3 b) { return length(a - b); }
HPRT_HD float max_comp(vec3 v) { return sel_max(v.x, sel_max(v.y, v.z)); }
HPRT_HD int max_dim(vec3 v) { return (v.x > v.y) ? ((v.x > v.z) ? 0 : 2) : ((v.y > v.z) ? 1 : 2); }
HPRT_HD vec3 vmin(vec3 a, vec3 b) { return vec3(sel_min(a.x, b.x), sel_min(a.y, b.y), sel_min(a.z, b.z)); }
HPRT_HD vec3 vmax(vec3 a, vec3 b) { return vec3(sel_max(a.x, b.x), sel_max(a.y, b.y), sel_max(a.z, b.z)); }
HPRT_HD vec3 face_forward(vec3 n, vec3 v) { return (dot(n, v) < 0.f) ? -n : n; }
// geometry.h:1360-1367
HPRT_HD void coordinate_system(vec3 v1, vec3 *v2, vec3 *v3) {
    if (fabsf(v1.x) > fabsf(v1.y)) *v2 = div_by(vec3(-v1.z, 0.f, v1.x), sqrtf(v1.x * v1.x + v1.z * v1.z));
    else *v2 = div_by(vec3(0.f, v1.z, -v1.y), sqrtf(v1.y * v1.y + v1.z * v1.z));
    *v3 = cross(v1, *v2);
}
// geometry.h:1782-1802
HPRT_HD vec3 offset_ray_origin(vec3 p, vec3 pErr, vec3 n, vec3 w) {
    float d = dot(vabs(n), pErr);
    vec3 off = d * n;
    if (dot(w, n) < 0) off = -off;
    vec3 po = p + off;
    if (off.x > 0) po.x = next_up(po.x); else if (off.x < 0) po.x = next_down(po.x);
    if (off.y > 0) po.y = next_up(po.y); else if (off.y < 0) po.y = next_down(po.y);
    if (off.z > 0) po.z = next_up(po.z); else if (off.z < 0) po.z = next_down(po.z);
    return po;
}

// ---------------------------------------------------------------------------
// Deterministic sin/cos/atan2/acos.  The reference calls glibc here.  sinf and
// cosf (the two the path calls per sample) are glibc's own algorithm restated
// further down and return its values bit for bit; double sin/cos, atan2f and
// acosf (sphere parametrisation, one normal-incidence branch) are evaluated
// from IEEE-exact double operations (Cody-Waite reduction + Taylor
// polynomials, error ~1e-16) and rounded once to float: correctly rounded for
// all but ~1e-8 of arguments, which is not always glibc's value.  DESIGN.md
// §5 states the measured difference against glibc.
// ---------------------------------------------------------------------------
HPRT_HD double k_sin(double r) {
    double z = r * r;
    double p = -8.22063524662432971696e-18;
    p = p * z + 2.81145725434552076320e-15;
    p = p * z + -7.64716373181981647590e-13;
    p = p * z + 1.60590438368216145994e-10;
    p = p * z + -2.50521083854417187751e-08;
    p = p * z + 2.75573192239858906526e-06;
    p = p * z + -1.98412698412698412698e-04;
    p = p * z + 8.33333333333333333333e-03;
    p = p * z + -1.66666666666666666667e-01;
    return r + r * (z * p);
}
HPRT_HD double k_cos(double r) {
    double z = r * r;
    double p = 4.11031762331216485848e-19;
    p = p * z + -1.56192069685862264622e-16;
    p = p * z + 4.77947733238738529744e-14;
    p = p * z + -1.14707455977297247139e-11;
    p = p * z + 2.08767569878680989792e-09;
    p = p * z + -2.75573192239858906526e-07;
    p = p * z + 2.48015873015873015873e-05;
    p = p * z + -1.38888888888888888889e-03;
    p = p * z + 4.16666666666666666667e-02;
    p = p * z + -5.00000000000000000000e-01;
    return 1.0 + z * p;
}
HPRT_HD int reduce_pio2(double x, double *r) {
    const double P1 = 1.57079632673412561417e+00, P2 = 6.07710050630396597660e-11;
    const double P3 = 2.02226624871116645580e-21, P3T = 8.47842766036889956997e-32;
    double fk = x * 6.36619772367581382433e-01;
    long long k = (long long)(fk + (fk >= 0 ? 0.5 : -0.5));
    double dk = (double)k;
    double t = x - dk * P1;
    t = t - dk * P2;
    t = t - dk * P3;
    t = t - dk * P3T;
    *r = t;
    return (int)(k & 3);
}
HPRT_HD double det_sin(double x) {
    double r; int q = reduce_pio2(x, &r);
    double s = k_sin(r), c = k_cos(r);
    return q == 0 ? s : (q == 1 ? c : (q == 2 ? -s : -c));
}
HPRT_HD double det_cos(double x) {
    double r; int q = reduce_pio2(x, &r);
    double s = k_sin(r), c = k_cos(r);
    return q == 0 ? c : (q == 1 ? -s : (q == 2 ? -c : s));
}
// sin and cos of one argument with a single reduction and one evaluation of each kernel:
// the same operations, hence the same two values, as det_sin(x) and det_cos(x).
HPRT_HD void det_sincos(double x, double *sn, double *cs) {
    double r; int q = reduce_pio2(x, &r);
    double s = k_sin(r), c = k_cos(r);
    *sn = q == 0 ? s : (q == 1 ? c : (q == 2 ? -s : -c));
    *cs = q == 0 ? c : (q == 1 ? -s : (q == 2 ? -c : s));
}
// sinf / cosf: glibc 2.35's algorithm (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h — what the reference's
// std::sin(float) / std::cos(float) run), restated so the device returns glibc's value bit for bit: double arithmetic on
// the float argument, a degree-7 / degree-8 polynomial after one fused reduction step by n*(pi/2).  The fused
// multiply-adds are explicit because x86-64 glibc runs its FMA build of these files; everything else in this header is
// compiled with contraction off.  The interval tests compare the top 12 bits of the float as glibc does (0.75, 2^-12, 120).
// oracle/orc_math.h holds the CPU restatement and the account of the exhaustive check against libm.  |x| >= 120 is not
// reached by the path (arguments are 2*pi*u and pi/4*ratio) and keeps the series above.
HPRT_HD unsigned det_abstop12(float x) { unsigned u; memcpy(&u, &x, 4); return (u >> 20) & 0x7ffu; }
HPRT_HD float det_sincos_poly(double x, double x2, bool negCos, int n) {
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double s1 = __builtin_fma(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7);
        double x7 = x3 * x2;
        double s = __builtin_fma(x3, -0x1.555545995a603p-3, x);
        return (float)__builtin_fma(x7, s1, s);
    }
    const double sg = negCos ? -1.0 : 1.0;       // the second table row: the cosine coefficients negated (exact)
    double x4 = x2 * x2;
    double c2 = __builtin_fma(x2, sg * 0x1.99343027bf8c3p-16, sg * -0x1.6c087e89a359dp-10);
    double c1 = __builtin_fma(x2, sg * 0x1.55553e1068f19p-5, sg * -0x1.ffffffd0c621cp-2);
    double x6 = x4 * x2;
    double c = __builtin_fma(x2, c1, sg);
    return (float)__builtin_fma(x6, c2, c);
}
HPRT_HD double det_sincos_reduce(double x, int *np) {
    double r = x * 0x1.45F306DC9C883p+23;
    int n = ((int)r + 0x800000) >> 24;
    *np = n;
    return __builtin_fma(-(double)n, 0x1.921FB54442D18p0, x);
}
HPRT_HD void det_sincosf(float y, float *sn, float *cs) {
    double x = y;
    const unsigned top = det_abstop12(y);
    if (top < 0x3f4u) {
        if (top < 0x398u) { *sn = y; *cs = 1.0f; return; }
        double x2 = x * x;
        *sn = det_sincos_poly(x, x2, false, 0);
        *cs = det_sincos_poly(x, x2, false, 1);
        return;
    }
    if (top < 0x42fu) {
        int n; x = det_sincos_reduce(x, &n);
        const int q = n & 3;
        const double xs = (q == 1 || q == 2) ? -x : x;
        const double x2 = x * x;
        const bool neg = (n & 2) != 0;
        *sn = det_sincos_poly(xs, x2, neg, n);
        *cs = det_sincos_poly(xs, x2, neg, n ^ 1);
        return;
    }
    double s, c; det_sincos((double)y, &s, &c); *sn = (float)s; *cs = (float)c;
}
HPRT_HD double det_atan_pos(double t) {
    double base = 0.0; bool inv = false;
    if (t > 1.0) { t = 1.0 / t; inv = true; }
    if (t > 0.41421356237309504880) { base = 0.78539816339744830962; t = (t - 1.0) / (t + 1.0); }
    double z = t * t;
    double p = 1.0 / 45.0;
    for (int n = 21; n >= 0; --n) {
        double c = 1.0 / (double)(2 * n + 1);
        p = c - z * p;
    }
    double a = base + t * p;
    return inv ? (1.57079632679489661923 - a) : a;
}
HPRT_HD double det_atan2(double y, double x) {
    if (x == 0.0) {
        if (y == 0.0) return 0.0;
        return y > 0 ? 1.57079632679489661923 : -1.57079632679489661923;
    }
    double a = det_atan_pos(fabs(y) / fabs(x));
    if (x < 0) a = 3.14159265358979323846 - a;
    return (y < 0) ? -a : a;
}
HPRT_HD double det_acos(double x) {
    if (x <= -1.0) return 3.14159265358979323846;
    if (x >= 1.0) return 0.0;
    return 2.0 * det_atan_pos(sqrt((1.0 - x) / (1.0 + x)));
}
// log(x) for finite x > 0: x = m * 2^e with m in [sqrt(1/2), sqrt(2)), log(m) = 2 atanh((m-1)/(m+1)) as a series
// (stands in for logf of Log2(), core/pbrt.h:328-331, like the functions above do for sinf / cosf)
HPRT_HD double det_log(double x) {
    if (!(x > 0.0)) return x == 0.0 ? -HUGE_VAL : (x - x) / (x - x);
    if (x > 1.7976931348623157e308) return x;
    unsigned long long bits; memcpy(&bits, &x, 8);
    int e = (int)((bits >> 52) & 0x7ffu);
    if (e == 0) { x *= 18014398509481984.0; memcpy(&bits, &x, 8); e = (int)((bits >> 52) & 0x7ffu) - 54; }
    e -= 1023;
    bits = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m; memcpy(&m, &bits, 8);
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0), z = s * s;
    double p = 1.0 / 27.0;
    for (int n = 12; n >= 0; --n) p = 1.0 / (double)(2 * n + 1) + z * p;
    return (double)e * 6.93147180369123816490e-01 + ((double)e * 1.90821492927058770002e-10 + 2.0 * s * p);
}

// logf: glibc 2.35's sysdeps/ieee754/flt-32/e_logf.c + e_logf_data.c (ARM optimized routines: 16-entry table of {1/c, log c},
// cubic in r = z/c - 1, double arithmetic) restated.  Equal to libm on all 2,139,095,039 positive finite floats, with and
// without fused multiply-adds (tools/debug/logf_exhaustive.c), so the plain form is used.
HPRT_HD float det_logf_glibc(float x) {
    const double T[16][2] = {
        {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2}, {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},
        {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3}, {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
        {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4}, {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5},
        {0x1p+0, 0x0p+0}, {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5}, {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
        {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3}, {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3}, {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},
        {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
    const double Ln2 = 0x1.62e42fefa39efp-1, A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    unsigned ix; memcpy(&ix, &x, 4);
    if (ix == 0x3f800000u) return 0.f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2u == 0u) return -HUGE_VALF;                                           // log(0) = -inf
        if (ix == 0x7f800000u) return x;                                                // log(inf) = inf
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return (x - x) / (x - x);     // negative or NaN
        const float xs = x * 0x1p23f;                                                   // subnormal: normalise
        memcpy(&ix, &xs, 4); ix -= 23u << 23;
    }
    const unsigned tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) % 16u), k = (int)tmp >> 23;
    const unsigned iz = ix - (tmp & (0x1ffu << 23));
    float zf; memcpy(&zf, &iz, 4);
    const double z = (double)zf;
    const double r = z * T[i][0] - 1.0;
    const double y0 = T[i][1] + (double)k * Ln2;
    const double r2 = r * r;
    double y = A1 * r + A2;
    y = A0 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}
HPRT_HD float det_logf(float x) { return det_logf_glibc(x); }
HPRT_HD float det_sinf(float x) { float s, c; det_sincosf(x, &s, &c); return s; }
HPRT_HD float det_cosf(float x) { float s, c; det_sincosf(x, &s, &c); return c; }
// acosf / atanf / atan2f: glibc 2.35's float routines (fdlibm; e_acosf.c, s_atanf.c, e_atan2f.c) restated — what the reference's
// std::acos(float) / std::atan2(float, float) run in shapes/sphere.cpp.  Float arithmetic, contraction off, correctly rounded
// divide and sqrt: the device returns glibc's values bit for bit (oracle/orc_math.h has the account of the exhaustive check).


HPRT_HD int det_f2i(float x) { int i; memcpy(&i, &x, 4); return i; }
HPRT_HD float det_i2f(int i) { float x; memcpy(&x, &i, 4); return x; }
// e_acosf.c
HPRT_HD float det_acosf_glibc(float x) {
    const float one = 1.0f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f,
                pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f,
                pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f,
                qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    const int hx = det_f2i(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;       // |x| == 1
    if (ix > 0x3f800000) return (x - x) / (x - x);                          // |x| > 1: NaN
    if (ix < 0x3f000000) {                                                  // |x| < 0.5
        if (ix <= 0x32800000) return pio2_hi + pio2_lo;
        const float z = x * x;
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (hx < 0) {                                                           // x < -0.5
        const float z = (one + x) * 0.5f;
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float s = sqrtf(z);
        const float r = p / q;
        const float w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    const float z = (one - x) * 0.5f;                                       // x > 0.5
    const float s = sqrtf(z);
    const float df = det_i2f(det_f2i(s) & (int)0xfffff000);
    const float c = (z - df * df) / (s + df);
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    const float w = r * s + c;
    return 2.0f * (df + w);
}
// s_atanf.c
HPRT_HD float det_atanf_glibc(float x) {
    const float one = 1.0f;
    const float hi0 = 4.6364760399e-01f, hi1 = 7.8539812565e-01f, hi2 = 9.8279368877e-01f, hi3 = 1.5707962513e+00f;
    const float lo0 = 5.0121582440e-09f, lo1 = 3.7748947079e-08f, lo2 = 3.4473217170e-08f, lo3 = 7.5497894159e-08f;
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    const int hx = det_f2i(x), ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {                                                 // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? hi3 + lo3 : -hi3 - lo3;
    }
    if (ix < 0x3ee00000) {                                                  // |x| < 0.4375
        if (ix < 0x31000000) return x;
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - one) / (2.0f + x); }
            else { id = 1; x = (x - one) / (x + one); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (one + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    const float z = x * x, w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    const float hi = id == 0 ? hi0 : id == 1 ? hi1 : id == 2 ? hi2 : hi3, lo = id == 0 ? lo0 : id == 1 ? lo1 : id == 2 ? lo2 : lo3;
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -r : r;
}
// e_atan2f.c (finite arguments; infinities do not occur on the path and return NaN here)
HPRT_HD float det_atan2f_glibc(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int hx = det_f2i(x), hy = det_f2i(y), ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return det_atanf_glibc(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000 || iy == 0x7f800000) return (x - x) / (x - x);
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = det_atanf_glibc(fabsf(y / x));
    switch (m) {
    case 0: return z;
    case 1: return det_i2f(det_f2i(z) ^ (int)0x80000000);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

HPRT_HD float det_atan2f(float y, float x) { return det_atan2f_glibc(y, x); }
HPRT_HD float det_acosf(float x) { return det_acosf_glibc(x); }

// RGB spectrum as three floats (core/spectrum.h RGBSpectrum)
struct rgb {
    float r, g, b;
    HPRT_HD rgb() : r(0.f), g(0.f), b(0.f) {}
    HPRT_HD explicit rgb(float v) : r(v), g(v), b(v) {}
    HPRT_HD rgb(float r_, float g_, float b_) : r(r_), g(g_), b(b_) {}
};
HPRT_HD rgb operator+(rgb a, rgb b) { return rgb(a.r + b.r, a.g + b.g, a.b + b.b); }
HPRT_HD rgb operator-(rgb a, rgb b) { return rgb(a.r - b.r, a.g - b.g, a.b - b.b); }
HPRT_HD rgb operator-(rgb a) { return rgb(-a.r, -a.g, -a.b); }
HPRT_HD rgb operator*(rgb a, rgb b) { return rgb(a.r * b.r, a.g * b.g, a.b * b.b); }
HPRT_HD rgb operator*(rgb a, float s) { return rgb(a.r * s, a.g * s, a.b * s); }
HPRT_HD rgb operator*(float s, rgb a) { return rgb(a.r * s, a.g * s, a.b * s); }
HPRT_HD rgb operator/(rgb a, float s) { return rgb(a.r / s, a.g / s, a.b / s); }   // spectrum.h:181-187: true divide
HPRT_HD bool is_black(rgb a) { return a.r == 0.f && a.g == 0.f && a.b == 0.f; }
HPRT_HD float max_value(rgb a) { float m = a.r; m = sel_max(m, a.g); m = sel_max(m, a.b); return m; }
HPRT_HD float luminance(rgb a) { return 0.212671f * a.r + 0.715160f * a.g + 0.072169f * a.b; }
HPRT_HD rgb clamp0(rgb a) {   // Spectrum::Clamp(0, Infinity)
    return rgb(clampf(a.r, 0.f, HPRT_INF), clampf(a.g, 0.f, HPRT_INF), clampf(a.b, 0.f, HPRT_INF));
}

// 4x4 matrix application, core/transform.h:220-347
struct mat4 { float m[4][4]; };
HPRT_HD vec3 xf_point(const mat4 &M, vec3 p) {
    float x = p.x, y = p.y, z = p.z;
    float xp = M.m[0][0] * x + M.m[0][1] * y + M.m[0][2] * z + M.m[0][3];
    float yp = M.m[1][0] * x + M.m[1][1] * y + M.m[1][2] * z + M.m[1][3];
    float zp = M.m[2][0] * x + M.m[2][1] * y + M.m[2][2] * z + M.m[2][3];
    float wp = M.m[3][0] * x + M.m[3][1] * y + M.m[3][2] * z + M.m[3][3];
    if (wp == 1.f) return vec3(xp, yp, zp);
    return div_by(vec3(xp, yp, zp), wp);
}
HPRT_HD vec3 xf_vector(const mat4 &M, vec3 v) {
    float x = v.x, y = v.y, z = v.z;
    return vec3(M.m[0][0] * x + M.m[0][1] * y + M.m[0][2] * z, M.m[1][0] * x + M.m[1][1] * y + M.m[1][2] * z,
                M.m[2][0] * x + M.m[2][1] * y + M.m[2][2] * z);
}
// normals use the transpose of the inverse: pass the inverse matrix
HPRT_HD vec3 xf_normal(const mat4 &Minv, vec3 n) {
    float x = n.x, y = n.y, z = n.z;
    return vec3(Minv.m[0][0] * x + Minv.m[1][0] * y + Minv.m[2][0] * z,
                Minv.m[0][1] * x + Minv.m[1][1] * y + Minv.m[2][1] * z,
                Minv.m[0][2] * x + Minv.m[1][2] * y + Minv.m[2][2] * z);
}
HPRT_HD vec3 xf_abs_row_sums(const mat4 &M, vec3 p, bool withTranslation) {
    float x = p.x, y = p.y, z = p.z;
    float sx = fabsf(M.m[0][0] * x) + fabsf(M.m[0][1] * y) + fabsf(M.m[0][2] * z);
    float sy = fabsf(M.m[1][0] * x) + fabsf(M.m[1][1] * y) + fabsf(M.m[1][2] * z);
    float sz = fabsf(M.m[2][0] * x) + fabsf(M.m[2][1] * y) + fabsf(M.m[2][2] * z);
    if (withTranslation) { sx = sx + fabsf(M.m[0][3]); sy = sy + fabsf(M.m[1][3]); sz = sz + fabsf(M.m[2][3]); }
    return vec3(sx, sy, sz);
}
// transform.h:277-296
HPRT_HD vec3 xf_point_err(const mat4 &M, vec3 p, vec3 *pErr) {
    *pErr = gamma_n(3) * xf_abs_row_sums(M, p, true);
    return xf_point(M, p);
}
// transform.h:298-328
HPRT_HD vec3 xf_point_err_in(const mat4 &M, vec3 p, vec3 e, vec3 *outErr) {
    float g3 = gamma_n(3);
    vec3 s = xf_abs_row_sums(M, p, true);
    outErr->x = (g3 + 1.0f) * (fabsf(M.m[0][0]) * e.x + fabsf(M.m[0][1]) * e.y + fabsf(M.m[0][2]) * e.z) + g3 * s.x;
    outErr->y = (g3 + 1.0f) * (fabsf(M.m[1][0]) * e.x + fabsf(M.m[1][1]) * e.y + fabsf(M.m[1][2]) * e.z) + g3 * s.y;
    outErr->z = (g3 + 1.0f) * (fabsf(M.m[2][0]) * e.x + fabsf(M.m[2][1]) * e.y + fabsf(M.m[2][2]) * e.z) + g3 * s.z;
    return xf_point(M, p);
}
// transform.h:330-347
HPRT_HD vec3 xf_vector_err(const mat4 &M, vec3 v, vec3 *vErr) {
    *vErr = gamma_n(3) * xf_abs_row_sums(M, v, false);
    return xf_vector(M, v);
}

}  // namespace hprt
