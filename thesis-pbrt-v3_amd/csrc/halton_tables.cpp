// hprt host side — see halton_tables.h.
#include "halton_tables.h"
#include <algorithm>

namespace hprt {
namespace {

struct Pcg32 {                         // core/rng.h:61-144, default-constructed stream
    uint64_t state = 0x853c49e6748fea9bULL, inc = 0xda3e39cb94b95bdbULL;
    uint32_t next() {
        uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xs >> rot) | (xs << ((~rot + 1u) & 31));
    }
    uint32_t bounded(uint32_t b) {     // UniformUInt32(b): rejection to kill modulo bias
        uint32_t threshold = (~b + 1u) % b;
        for (;;) { uint32_t r = next(); if (r >= threshold) return r % b; }
    }
};

int64_t posMod(int64_t a, int64_t b) { int64_t r = a - (a / b) * b; return r < 0 ? r + b : r; }
void extGcd(uint64_t a, uint64_t b, int64_t *x, int64_t *y) {   // samplers/halton.cpp:52-62
    if (b == 0) { *x = 1; *y = 0; return; }
    int64_t d = a / b, xp, yp;
    extGcd(b, a % b, &xp, &yp);
    *x = yp; *y = xp - (d * yp);
}
uint64_t invRadical(int base, uint64_t inverse, int nDigits) {  // lowdiscrepancy.h:82-91
    uint64_t index = 0;
    for (int i = 0; i < nDigits; ++i) { uint64_t digit = inverse % base; inverse /= base; index = index * base + digit; }
    return index;
}

}  // namespace

const std::vector<int> &PrimeTable() {
    static const std::vector<int> table = [] {
        std::vector<int> p; std::vector<char> composite(8000, 0);   // the 1000th prime is 7919
        for (int i = 2; i < 8000 && (int)p.size() < kPrimeTableSize; ++i) {
            if (composite[i]) continue;
            p.push_back(i);
            for (int j = i * i; j < 8000; j += i) composite[j] = 1;
        }
        return p;
    }();
    return table;
}
const std::vector<int> &PrimeSumTable() {
    static const std::vector<int> sums = [] {
        std::vector<int> s; int acc = 0;
        for (int p : PrimeTable()) { s.push_back(acc); acc += p; }
        return s;
    }();
    return sums;
}
const std::vector<uint16_t> &HaltonPermutations() {
    static const std::vector<uint16_t> perms = [] {
        std::vector<uint16_t> all;
        Pcg32 rng;
        for (int prime : PrimeTable()) {
            size_t base = all.size();
            for (int j = 0; j < prime; ++j) all.push_back((uint16_t)j);
            for (int i = 0; i < prime; ++i) {          // Shuffle(p, count, 1, rng)
                int other = i + (int)rng.bounded((uint32_t)(prime - i));
                std::swap(all[base + i], all[base + other]);
            }
        }
        return all;
    }();
    return perms;
}

HaltonLayout MakeHaltonLayout(int resX, int resY) {
    const int kMaxResolution = 128;
    HaltonLayout h;
    int res[2] = {resX, resY};
    for (int i = 0; i < 2; ++i) {
        int base = (i == 0) ? 2 : 3, scale = 1, exp = 0;
        while (scale < std::min(res[i], kMaxResolution)) { scale *= base; ++exp; }
        h.baseScales[i] = scale; h.baseExponents[i] = exp;
    }
    h.sampleStride = h.baseScales[0] * h.baseScales[1];
    int64_t x, y;
    extGcd(h.baseScales[1], h.baseScales[0], &x, &y); h.multInverse[0] = (int)posMod(x, h.baseScales[0]);
    extGcd(h.baseScales[0], h.baseScales[1], &x, &y); h.multInverse[1] = (int)posMod(x, h.baseScales[1]);
    return h;
}
int64_t HaltonPixelOffset(const HaltonLayout &h, int px, int py) {
    const int kMaxResolution = 128;
    uint64_t off = 0;
    if (h.sampleStride > 1) {
        int pm[2] = {(int)posMod(px, kMaxResolution), (int)posMod(py, kMaxResolution)};
        for (int i = 0; i < 2; ++i) {
            uint64_t dimOffset = invRadical(i == 0 ? 2 : 3, (uint64_t)pm[i], h.baseExponents[i]);
            off += dimOffset * (uint64_t)(h.sampleStride / h.baseScales[i]) * (uint64_t)h.multInverse[i];
        }
        off %= (uint64_t)h.sampleStride;
    }
    return (int64_t)off;
}

}  // namespace hprt
