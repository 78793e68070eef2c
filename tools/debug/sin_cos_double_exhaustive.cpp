// Exhaustive check of the restated glibc double sin / cos — the product's (hprt_math.h det_sincos_glibc_d) and the oracle's
// (orc_math.h det::sin_glibc_d / cos_glibc_d) — against the libm of the machine it runs on, over every float argument in
// [0, 2 pi): the whole domain of `cos(phi)` / `sin(phi)` in TrowbridgeReitzSample11 (core/microfacet.cpp:243-245).
//   g++ -O2 -fopenmp -ffp-contract=off -fno-builtin -mfma -std=c++17 -I. tools/debug/sin_cos_double_exhaustive.cpp -lm
// glibc 2.35 on a CPU with FMA: 1,086,918,619 arguments, 0 mismatches for all four functions.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "oracle/orc_math.h"
#include "thesis-pbrt-v3_amd/csrc/hprt_math.h"
namespace orc { bool g_use_libm = false; }
int main() {
    long bad[4] = {0, 0, 0, 0}, n = 0;
    const uint32_t last = 0x40c90fdau;      // (float)(6.28318530718 * 0x1.fffffep-1)
#pragma omp parallel for reduction(+ : bad[:4], n)
    for (int64_t i = 0; i <= (int64_t)last; ++i) {
        uint32_t u = (uint32_t)i; float x; memcpy(&x, &u, 4);
        volatile double xx = x;
        const double s = ::sin(xx), c = ::cos(xx);
        const double so = orc::det::sin_glibc_d((double)x), co = orc::det::cos_glibc_d((double)x);
        double sp, cp; hprt::det_sincos_glibc_d((double)x, &sp, &cp);
        ++n;
        bad[0] += memcmp(&s, &so, 8) != 0; bad[1] += memcmp(&c, &co, 8) != 0;
        bad[2] += memcmp(&s, &sp, 8) != 0; bad[3] += memcmp(&c, &cp, 8) != 0;
    }
    printf("%ld arguments: oracle sin %ld, cos %ld mismatches; product sin %ld, cos %ld mismatches\n", n, bad[0], bad[1], bad[2], bad[3]);
    return bad[0] + bad[1] + bad[2] + bad[3] != 0;
}
