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
// sin and cos of a double: glibc 2.35's sysdeps/ieee754/dbl-64/s_sin.c (__sin / __cos, the IBM Accurate Mathematical Library
// routines: a 1/128-spaced table of sin and cos as double-doubles, short polynomials around the table point, a two-constant
// reduction by pi/2) restated — what the reference's unqualified `cos(phi)` / `sin(phi)` in TrowbridgeReitzSample11
// (core/microfacet.cpp:243-245) run: the call resolves to ::cos(double).  x86-64 glibc selects its FMA build of that file on
// CPUs that have FMA, so the multiply-adds below are fused where that build fuses them.  Equal to libm's sin and cos on
// every float argument in [0, 2 pi) — all 1,086,918,619 of them, the whole domain of that call site
// (tools/debug/sin_cos_double_exhaustive.cpp; the table is recomputed by tools/debug/sincos_table.py).  Valid for |x| < 1e8.
HPRT_HD double det_bits2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
HPRT_HD void det_sincostab(int k, double *sn, double *ssn, double *cs, double *ccs) {
    const double T[112][4] = {
    {0.0, 0.0, 0x1.0000000000000p+0, 0.0},
    {0x1.fffeaaaaeeeefp-8, -0x1.e45e2ec67b77cp-62, 0x1.fffc000155552p-1, 0x1.f4a01a0196daep-55},
    {0x1.fffaaaaeeeed5p-7, -0x1.2ab639a9f0776p-63, 0x1.fff000155549fp-1, 0x1.28a28a03a5ef3p-55},
    {0x1.7ff7001033255p-6, 0x1.efe2b51527336p-64, 0x1.ffdc006bff7e6p-1, 0x1.ae6dae86977bdp-55},
    {0x1.ffeaaaeeee86fp-6, -0x1.cd406fb224ae2p-60, 0x1.ffc00155527d3p-1, -0x1.3b54492d89b5bp-55},
    {0x1.3feb2b12d45d5p-5, 0x1.4ec54203d1c11p-60, 0x1.ff9c03414a7bap-1, 0x1.991f4be6c59bfp-57},
    {0x1.7fdc01032fba9p-5, -0x1.599bdf46e997ap-59, 0x1.ff7006bfdf99fp-1, -0x1.8b3b560648d5fp-56},
    {0x1.bfc6d78586dacp-5, 0x1.8e4fd03dbf236p-62, 0x1.ff3c0c8103a31p-1, 0x1.4856dbddc0e66p-56},
    {0x1.ffaaaeeed4edbp-5, -0x1.2d16d32684b69p-59, 0x1.ff0015549f4d3p-1, 0x1.328387b99426fp-55},
    {0x1.1fc343d808befp-4, -0x1.f3d32e6f3be4fp-58, 0x1.febc222a8ef9fp-1, 0x1.7934934f54c77p-58},
    {0x1.3facb12d1755bp-4, -0x1.921915299468bp-58, 0x1.fe7034129ef6fp-1, -0x1.cbf4337c96f97p-57},
    {0x1.5f911fd10b737p-4, -0x1.0184f02be9102p-58, 0x1.fe1c4c3c873ebp-1, -0x1.5a9c9057c4a02p-60},
    {0x1.7f701032550e4p-4, 0x1.afc2d1800501ap-60, 0x1.fdc06bf7e6b9bp-1, 0x1.31902b535f8dbp-55},
    {0x1.9f4902d55d1f9p-4, 0x1.2696d7eac1dc1p-58, 0x1.fd5c94b43e000p-1, -0x1.2e768cb4f92f9p-57},
    {0x1.bf1b78568391dp-4, 0x1.e91841dea4cc8p-58, 0x1.fcf0c800e99b1p-1, 0x1.ea3d786d186acp-57},
    {0x1.dee6f16c1cce6p-4, -0x1.50f8e2fb71673p-59, 0x1.fc7d078d1bc88p-1, 0x1.075d2447db685p-55},
    {0x1.feaaeee86ee36p-4, -0x1.afcb2bcc6f03bp-59, 0x1.fc015527d5bd3p-1, 0x1.b68f35094efb8p-55},
    {0x1.0f3378ddd71d1p-3, 0x1.d8468724f0f9ep-57, 0x1.fb7db2bfe0695p-1, 0x1.21dadf4f65ab1p-55},
    {0x1.1f0d3d7afceafp-3, -0x1.6ef95099769a5p-57, 0x1.faf22263c4bd3p-1, -0x1.52ace133a2769p-58},
    {0x1.2ee285e4ab88fp-3, -0x1.e4d0f05dee058p-57, 0x1.fa5ea641c36f2p-1, 0x1.04da6ed17cc7cp-59},
    {0x1.3eb312c5d66cbp-3, 0x1.47d666b66cb91p-57, 0x1.f9c340a7cc428p-1, 0x1.c5b6b063b7462p-55},
    {0x1.4e7ea4dc5f27bp-3, 0x1.949db2ac072fcp-58, 0x1.f91ff40374d01p-1, -0x1.7d03f4d3a9e4cp-57},
    {0x1.5e44fcfa126f3p-3, -0x1.6f443063f89b6p-57, 0x1.f874c2e1eecf6p-1, -0x1.c6514e1332b16p-55},
    {0x1.6e05dc05a4d4cp-3, -0x1.32c5c8b81c919p-66, 0x1.f7c1afeffde24p-1, -0x1.8f55bc47540b1p-56},
    {0x1.7dc102fbaf2b5p-3, 0x1.5ab50e23c97c3p-59, 0x1.f706bdf9ece1cp-1, -0x1.698c80c36dcb4p-55},
    {0x1.8d7632efaa944p-3, -0x1.20fa262cbb953p-57, 0x1.f643efeb82acdp-1, 0x1.6b00ac1fe28acp-56},
    {0x1.9d252d0cec312p-3, 0x1.9c43d80b1137dp-58, 0x1.f57948cff6797p-1, 0x1.e3a0d3e03b1d4p-57},
    {0x1.accdb297a0765p-3, -0x1.9883b57d6cdeap-58, 0x1.f4a6cbd1e3a79p-1, 0x1.13df0edaebb57p-55},
    {0x1.bc6f84edc6199p-3, 0x1.9c1a56a7b0cabp-57, 0x1.f3cc7c3b3d16ep-1, -0x1.21a3ad28a3494p-57},
    {0x1.cc0a6588289a3p-3, -0x1.868d09bc87c6bp-57, 0x1.f2ea5d753ffedp-1, 0x1.cc4215f56d583p-55},
    {0x1.db9e15fb5a5d0p-3, -0x1.32e20d6cc6fc2p-57, 0x1.f20073086649fp-1, 0x1.b940416c1984bp-56},
    {0x1.eb2a57f8ae5a3p-3, -0x1.0be06af572cebp-57, 0x1.f10ec09c5873bp-1, 0x1.d9072762c1283p-55},
    {0x1.faaeed4f31577p-3, -0x1.15d88508e32b8p-57, 0x1.f01549f7deea1p-1, 0x1.d3c1e99e5cafdp-55},
    {0x1.0515cbf65155cp-2, -0x1.9b8c29dfd8ec7p-56, 0x1.ef141300d2f26p-1, -0x1.2aa1b08ded372p-55},
    {0x1.0cd00cef36436p-2, -0x1.9fb0a0c93e2b4p-56, 0x1.ee0b1fbc0f11cp-1, -0x1.bfd2380bbc3b1p-59},
    {0x1.14861aa94ddebp-2, -0x1.be881b5b615a4p-57, 0x1.ecfa744d5efa1p-1, -0x1.56d0a4af541d0p-58},
    {0x1.1c37d64c6b876p-2, 0x1.46076fe0dcff4p-56, 0x1.ebe214f76efa8p-1, -0x1.02f9f12ba543ep-55},
    {0x1.23e52111aaf36p-2, -0x1.4f080334eff18p-56, 0x1.eac2061bbaf4fp-1, 0x1.2c1d53e94658dp-57},
    {0x1.2b8ddc43eb49fp-2, 0x1.1553899f2d807p-57, 0x1.e99a4c3a7cd83p-1, -0x1.2264b1bc53ce8p-55},
    {0x1.3331e94049f87p-2, 0x1.e0cb6b40c302cp-56, 0x1.e86aebf29a9edp-1, 0x1.9397afdbb58a7p-55},
    {0x1.3ad129769d3d8p-2, 0x1.03d550487839ap-63, 0x1.e733ea0193d40p-1, -0x1.6428b3546ce13p-55},
    {0x1.426b7e69ee697p-2, -0x1.f09c75705c59fp-56, 0x1.e5f54b436e9d0p-1, 0x1.7eb0fd02fc8bcp-55},
    {0x1.4a00c9b0f3d20p-2, 0x1.823ba6bb08eadp-56, 0x1.e4af14b2a449cp-1, -0x1.68ca02e8a6833p-55},
    {0x1.5190ecf68a77ap-2, 0x1.b357155eef0f3p-56, 0x1.e3614b680d6a5p-1, -0x1.27793aa015237p-56},
    {0x1.591bc9fa2f597p-2, 0x1.7c74bac3fe0cbp-57, 0x1.e20bf49acd6c1p-1, -0x1.660aec7ef636bp-58},
    {0x1.60a1429078775p-2, 0x1.b1fd80ba89133p-58, 0x1.e0af15a03dbcep-1, 0x1.fe8e702771ae6p-58},
    {0x1.682138a38d7f7p-2, -0x1.d889202444aadp-56, 0x1.df4ab3ebd875ep-1, -0x1.e2d8a7e6736c4p-55},
    {0x1.6f9b8e33a0255p-2, 0x1.42bc14ee9da0dp-56, 0x1.ddded50f228d6p-1, -0x1.e80c8d42ba2bfp-57},
    {0x1.7710255764214p-2, -0x1.6ead7314bb6cep-57, 0x1.dc6b7eb995912p-1, 0x1.4b364776dcd35p-58},
    {0x1.7e7ee03c86d4ep-2, -0x1.b63bcdabf5af2p-56, 0x1.daf0b6b888e83p-1, 0x1.a249e2b5e5ceap-55},
    {0x1.85e7a12826949p-2, 0x1.8a40e9b5face0p-56, 0x1.d96e82f71a9dcp-1, 0x1.ff61bd5d2039dp-55},
    {0x1.8d4a4a774992fp-2, 0x1.44a02ea766326p-56, 0x1.d7e4e97e17b4ap-1, -0x1.3b770352bed94p-57},
    {0x1.94a6be9f546c5p-2, -0x1.69ce13e683f58p-56, 0x1.d653f073e4040p-1, -0x1.76236434bec37p-55},
    {0x1.9bfce02e80510p-2, 0x1.09e39a320b0a4p-56, 0x1.d4bb9e1c619e0p-1, 0x1.f34bb77858f61p-55},
    {0x1.a34c91cc50ccap-2, -0x1.a310e3b50cecdp-58, 0x1.d31bf8d8d7c06p-1, 0x1.e60dd3089cbddp-56},
    {0x1.aa95b63a09277p-2, -0x1.6293eb13c0381p-57, 0x1.d1750727d94f0p-1, 0x1.0d52b1ec1a48ep-55},
    {0x1.b1d8305321617p-2, -0x1.ae242cb99f519p-56, 0x1.cfc6cfa52ad9fp-1, 0x1.8b5b5508f2a0dp-55},
    {0x1.b913e30dbac43p-2, -0x1.e38ad2f6c3ff1p-56, 0x1.ce115909a82e5p-1, 0x1.1f139bb31109ap-55},
    {0x1.c048b17b140a3p-2, 0x1.19fe6757e9fa7p-57, 0x1.cc54aa2b2972ep-1, 0x1.4ee162ba83a98p-57},
    {0x1.c7767ec7fd19ep-2, -0x1.eb14d1a3d5826p-58, 0x1.ca90c9fc67d0bp-1, -0x1.46a81485e3462p-57},
    {0x1.ce9d2e3d4a51fp-2, -0x1.2fc8a12dae298p-57, 0x1.c8c5bf8ce1a84p-1, 0x1.ab3d1a1590123p-56},
    {0x1.d5bca34047661p-2, 0x1.28a44a75fc29cp-56, 0x1.c6f39208be53bp-1, -0x1.741dbfbaadb42p-55},
    {0x1.dcd4c15329c9ap-2, 0x1.0d4c6e171fd9ap-56, 0x1.c51a48b8b175ep-1, -0x1.1bbb43b9aa880p-57},
    {0x1.e3e56c1582a69p-2, -0x1.0a4821099f88fp-58, 0x1.c339eb01ddd81p-1, -0x1.caaf5ee82c5c0p-55},
    {0x1.eaee8744b05f0p-2, -0x1.789b43c9b027dp-58, 0x1.c1528065b7d50p-1, -0x1.892111312e828p-55},
    {0x1.f1eff6bc4f97bp-2, 0x1.17212f8a7525cp-56, 0x1.bf641081e7536p-1, 0x1.b7bd71628a9a1p-55},
    {0x1.f8e99e76abc97p-2, 0x1.9d950af2d00a3p-58, 0x1.bd6ea310294f5p-1, 0x1.31bbcc88c109dp-56},
    {0x1.ffdb628d2f57ap-2, 0x1.f4a992e905b6ap-57, 0x1.bb723fe630f32p-1, 0x1.72bd2452d0a39p-56},
    {0x1.0362939c69955p-1, -0x1.2d8cd78397b01p-55, 0x1.b96eeef58840ep-1, 0x1.45a3cc78fade0p-58},
    {0x1.06d3686946e5bp-1, 0x1.3f5ae4538ff1bp-55, 0x1.b764b84b704c2p-1, -0x1.f5848c21b389bp-55},
    {0x1.0a4021e9e1001p-1, -0x1.6f643a13914f6p-55, 0x1.b553a410c104ep-1, 0x1.8ff7947027a15p-58},
    {0x1.0da8b26b5672ep-1, -0x1.a58def0bee909p-55, 0x1.b33bba89c8948p-1, 0x1.ea6a51d1f6ca9p-55},
    {0x1.110d0c4b69c3bp-1, 0x1.d918998809981p-55, 0x1.b11d04162a4c6p-1, 0x1.1dd561efbc0c2p-56},
    {0x1.146d21f8b7f82p-1, 0x1.bf9535e2739a8p-56, 0x1.aef78930bd275p-1, -0x1.f836279746f94p-56},
    {0x1.17c8e5f2eedb0p-1, 0x1.35e57102e2488p-57, 0x1.accb526f69de5p-1, 0x1.8fb6a8dd6b6ccp-55},
    {0x1.1b204acb02fddp-1, -0x1.f190c70cbb5fep-58, 0x1.aa98688308913p-1, -0x1.b83d607cd5072p-63},
    {0x1.1e7343236574cp-1, 0x1.22a3fa4f41d5ap-56, 0x1.a85ed4373e02dp-1, 0x1.9be06385ec792p-57},
    {0x1.21c1c1b0394cfp-1, 0x1.e5b324b23aa31p-58, 0x1.a61e9e72586afp-1, 0x1.58330e2fd453fp-55},
    {0x1.250bb93788bbbp-1, 0x1.ea3d02457bccep-56, 0x1.a3d7d0352bdcfp-1, -0x1.68dbaeca19669p-55},
    {0x1.28511c917a067p-1, -0x1.01df1d9a16b70p-55, 0x1.a18a729aee445p-1, 0x1.95e25736c0357p-60},
    {0x1.2b91dea88421ep-1, -0x1.fa371db216ab0p-55, 0x1.9f368ed912f85p-1, -0x1.1d200c5791606p-55},
    {0x1.2ecdf279a3082p-1, 0x1.d3557e0e7e37ep-55, 0x1.9cdc2e3f25e5cp-1, 0x1.3f99112993f62p-55},
    {0x1.32054b148bc4fp-1, 0x1.f6b42095a135bp-55, 0x1.9a7b5a36a6514p-1, 0x1.722cfcc9fa7a9p-55},
    {0x1.3537db9be0367p-1, 0x1.b327e7af040f0p-57, 0x1.98141c42e1310p-1, 0x1.d1ff80488f08dp-55},
    {0x1.386597456282bp-1, -0x1.10fada93b07a8p-56, 0x1.95a67e00cb1fdp-1, -0x1.0befda21f862dp-55},
    {0x1.3b8e715a2840ap-1, -0x1.97653a7d2f07ap-56, 0x1.93328926d9e92p-1, -0x1.bb77003600cdap-55},
    {0x1.3eb25d36cd53ap-1, -0x1.be570e1570fc0p-58, 0x1.90b84784ddaf7p-1, -0x1.0feb10ab93b87p-56},
    {0x1.41d14e4ba6790p-1, 0x1.4608fd287ecf5p-55, 0x1.8e37c303d9ad1p-1, -0x1.463a4b53d4bf8p-57},
    {0x1.44eb381cf386bp-1, -0x1.3ed6c1e6a5505p-55, 0x1.8bb105a5dc900p-1, 0x1.863e03e9474c1p-55},
    {0x1.48000e431159fp-1, -0x1.b194a7463ed10p-55, 0x1.89241985d871fp-1, 0x1.c48d9c413ed84p-55},
    {0x1.4b0fc46aab761p-1, 0x1.0da05738cc59cp-61, 0x1.869108d77a6c6p-1, 0x1.338ffe2bfe9ddp-56},
    {0x1.4e1a4e54ed51bp-1, -0x1.a492f89b7c76ap-55, 0x1.83f7dde701ca0p-1, -0x1.152cf609bc6e8p-59},
    {0x1.511f9fd7b351cp-1, -0x1.5c0e861c48831p-55, 0x1.8158a31916d5dp-1, -0x1.de8b90b8228dep-57},
    {0x1.541facddbb724p-1, 0x1.232c28520d391p-56, 0x1.7eb362eaa1488p-1, 0x1.a1d65a4a5959fp-58},
    {0x1.571a6966d59b3p-1, 0x1.c843b4d0fb197p-58, 0x1.7c0827f09e54fp-1, -0x1.c73d6d72aee68p-57},
    {0x1.5a0fc98813a12p-1, -0x1.d82e2b7d4227bp-55, 0x1.7956fcd7f6543p-1, -0x1.ab276e9d45ae4p-55},
    {0x1.5cffc16bf8f0dp-1, 0x1.96cb370eb578ap-55, 0x1.769fec655211fp-1, -0x1.827d5cf8c68c5p-57},
    {0x1.5fea4552a9e57p-1, 0x1.0b6cef7ee20b7p-55, 0x1.73e30174efba1p-1, -0x1.5d3ae3d94ad5fp-57},
    {0x1.62cf49921ac79p-1, -0x1.edd9855b6241ap-55, 0x1.712046fa77678p-1, 0x1.425b0a5029c81p-55},
    {0x1.65aec2963e755p-1, 0x1.126f96b71053cp-55, 0x1.6e57c800cf55ep-1, 0x1.60286dedbd0a6p-55},
    {0x1.6888a4e134b2fp-1, -0x1.6b7d37644d5e6p-55, 0x1.6b898fa9efb5dp-1, 0x1.15ac786ccf4b2p-56},
    {0x1.6b5ce50b7821ap-1, -0x1.5d5158f702e0fp-57, 0x1.68b5a92eb6253p-1, -0x1.9a91ad985f89cp-55},
    {0x1.6e2b77c40bde1p-1, -0x1.0e729857fad53p-56, 0x1.65dc1fdeb8cbap-1, -0x1.97c1b47337c77p-58},
    {0x1.70f451d0a8c40p-1, 0x1.97ede3885770dp-57, 0x1.62fcff20191c7p-1, 0x1.d9143895756efp-57},
    {0x1.73b7680dea578p-1, -0x1.2248306dc12a2p-56, 0x1.6018526f563dfp-1, 0x1.46ca5e0e432d0p-55},
    {0x1.7674af6f7b524p-1, 0x1.e9d3f94ac84a8p-56, 0x1.5d2e255f1f17ap-1, 0x1.0314104c8892bp-55},
    {0x1.792c1d0041d52p-1, -0x1.abf05eeb354ebp-55, 0x1.5a3e839824077p-1, 0x1.428aa2759be62p-55},
    {0x1.7bdda5e28b3c2p-1, 0x1.ad1197ccd0392p-59, 0x1.574978d8e83f2p-1, 0x1.f4714af282d23p-55},
    {0x1.7e893f5037959p-1, 0x1.0eefbaa650c4cp-55, 0x1.544f10f592ca5p-1, -0x1.e7ae8e6c7a62fp-55},
    {0x1.812ede9ae4ba4p-1, -0x1.7830adf402ddap-55, 0x1.514f57d7bf3dap-1, 0x1.47a108073c259p-56},
    {0x1.83ce792c1906ep-1, -0x1.f3899682b4a7dp-56, 0x1.4e4a597e4e10ep-1, 0x1.ccd992849f6c8p-56},
    {0x1.866804856db62p-1, 0x1.407b4e7476623p-57, 0x1.4b4021fd34a33p-1, -0x1.ee903cecc18cbp-55},
    };
    *sn = T[k][0]; *ssn = T[k][1]; *cs = T[k][2]; *ccs = T[k][3];
}
HPRT_HD double det_glibc_do_sin(double x, double dx) {
    const double sn3 = -0x1.5555555555515p-3, sn5 = 0x1.11110e829872fp-7, cs2 = 0.5, cs4 = -0x1.5555555555535p-5, cs6 = 0x1.6c16bedd9e239p-10;
    const double big = 0x1.8p45;
    const double xold = x;
    if (fabs(x) < 0.126) {                                   // TAYLOR_SIN
        const double s1 = -0x1.5555555555555p-3, s2 = 0x1.1111111110ecep-7, s3 = -0x1.a01a019db08b8p-13, s4 = 0x1.71de27b9a7ed9p-19, s5 = -0x1.addffc2fcdf59p-26;
        const double xx = x * x;
        const double p = __builtin_fma(__builtin_fma(__builtin_fma(__builtin_fma(s5, xx, s4), xx, s3), xx, s2), xx, s1);
        const double t = __builtin_fma(__builtin_fma(p, x, -0.5 * dx), xx, dx);
        return x + t;
    }
    if (x <= 0) dx = -dx;
    const double ux = big + fabs(x);
    uint64_t ub; memcpy(&ub, &ux, 8);
    x = fabs(x) - (ux - big);
    const double xx = x * x;
    const double s = x + __builtin_fma(x * xx, __builtin_fma(xx, sn5, sn3), dx);
    const double c = __builtin_fma(x, dx, xx * __builtin_fma(xx, __builtin_fma(xx, cs6, cs4), cs2));
    double sn, ssn, cs, ccs; det_sincostab((int)(uint32_t)ub, &sn, &ssn, &cs, &ccs);
    const double cor = __builtin_fma(cs, s, __builtin_fma(-sn, c, __builtin_fma(s, ccs, ssn)));
    return copysign(sn + cor, xold);
}
HPRT_HD double det_glibc_do_cos(double x, double dx) {
    const double sn3 = -0x1.5555555555515p-3, sn5 = 0x1.11110e829872fp-7, cs2 = 0.5, cs4 = -0x1.5555555555535p-5, cs6 = 0x1.6c16bedd9e239p-10;
    const double big = 0x1.8p45;
    if (x < 0) dx = -dx;
    const double ux = big + fabs(x);
    uint64_t ub; memcpy(&ub, &ux, 8);
    x = fabs(x) - (ux - big) + dx;
    const double xx = x * x;
    const double s = __builtin_fma(x * xx, __builtin_fma(xx, sn5, sn3), x);
    const double c = xx * __builtin_fma(xx, __builtin_fma(xx, cs6, cs4), cs2);
    double sn, ssn, cs, ccs; det_sincostab((int)(uint32_t)ub, &sn, &ssn, &cs, &ccs);
    const double cor = __builtin_fma(-sn, s, __builtin_fma(-cs, c, __builtin_fma(-s, ssn, ccs)));
    return cs + cor;
}
HPRT_HD int det_glibc_reduce_sincos(double x, double *a, double *da) {
    const double toint = 0x1.8p52, hpinv = det_bits2d(0x3FE45F306DC9C883ull), mp1 = det_bits2d(0x3FF921FB58000000ull), mp2 = det_bits2d(0xBE4DDE973C000000ull),
                 pp3 = det_bits2d(0xBC8CB3B398000000ull), pp4 = det_bits2d(0xBACD747F23E32ED7ull);
    const double t = __builtin_fma(x, hpinv, toint), xn = t - toint;
    uint64_t vb; memcpy(&vb, &t, 8);
    const double y = __builtin_fma(-xn, mp2, __builtin_fma(-xn, mp1, x));
    double t1 = xn * pp3;
    const double t2 = y - t1;
    double db = (y - t2) - t1;
    t1 = xn * pp4;
    const double b = t2 - t1;
    db += (t2 - b) - t1;
    *a = b; *da = db;
    return (int)(vb & 3);
}
HPRT_HD void det_sincos_glibc_d(double x, double *sn, double *cs) {
    const double hp0 = det_bits2d(0x3FF921FB54442D18ull), hp1 = det_bits2d(0x3C91A62633145C07ull);
    uint64_t u; memcpy(&u, &x, 8);
    const uint32_t k = (uint32_t)(u >> 32) & 0x7fffffffu;
    if (k >= 0x400368fdu) {                                  // 2.426265 <= |x| (< 105414350): one reduction serves both
        double a, da;
        const int n = det_glibc_reduce_sincos(x, &a, &da);
        const double rs = det_glibc_do_sin(a, da), rc = det_glibc_do_cos(a, da);
        const double vs = (n & 1) ? rc : rs, vc = ((n + 1) & 1) ? rc : rs;
        *sn = (n & 2) ? -vs : vs;
        *cs = ((n + 1) & 2) ? -vc : vc;
        return;
    }
    if (k < 0x3e500000u) *sn = x;                            // |x| < 2^-26
    else if (k < 0x3feb6000u) *sn = det_glibc_do_sin(x, 0);   // |x| < 0.855469
    else *sn = copysign(det_glibc_do_cos(hp0 - fabs(x), hp1), x);
    if (k < 0x3e400000u) *cs = 1.0;                          // |x| < 2^-27
    else if (k < 0x3feb6000u) *cs = det_glibc_do_cos(x, 0);
    else {
        const double y = hp0 - fabs(x), a = y + hp1, da = (y - a) + hp1;
        *cs = det_glibc_do_sin(a, da);
    }
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
