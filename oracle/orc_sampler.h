// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
// Halton sampler restatement: core/rng.h, core/lowdiscrepancy.{h,cpp},
// core/sampling.h:151-157, samplers/halton.cpp, core/sampler.cpp.
#pragma once
#include <vector>
#include "orc_math.h"

namespace orc {

// core/rng.h:61-144 (PCG32)
struct RNG {
    uint64_t state, inc;
    RNG() : state(0x853c49e6748fea9bULL), inc(0xda3e39cb94b95bdbULL) {}
    explicit RNG(uint64_t sequenceIndex) { SetSequence(sequenceIndex); }
    void SetSequence(uint64_t initseq) {       // core/rng.h:130-136
        state = 0u;
        inc = (initseq << 1u) | 1u;
        UniformUInt32();
        state += 0x853c49e6748fea9bULL;
        UniformUInt32();
    }
    Float UniformFloat() { return smin(OneMinusEpsilon, Float(UniformUInt32() * 0x1p-32f)); }   // core/rng.h:78-85
    uint32_t UniformUInt32() {
        uint64_t oldstate = state;
        state = oldstate * 0x5851f42d4c957f2dULL + inc;
        uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
        uint32_t rot = (uint32_t)(oldstate >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
    }
    uint32_t UniformUInt32(uint32_t b) {
        uint32_t threshold = (~b + 1u) % b;
        while (true) {
            uint32_t r = UniformUInt32();
            if (r >= threshold) return r % b;
        }
    }
};

static const int PrimeTableSize = 1000;
// core/lowdiscrepancy.cpp:40-122 holds a literal table of the first 1000
// primes; the oracle sieves them (same values by definition).
inline const std::vector<int> &Primes() {
    static const std::vector<int> table = [] {   // thread-safe one-time initialisation
        std::vector<int> p;
        const int N = 8000;  // 1000th prime is 7919
        std::vector<char> comp(N, 0);
        for (int i = 2; i < N && (int)p.size() < PrimeTableSize; ++i) {
            if (!comp[i]) {
                p.push_back(i);
                for (int j = i * i; j < N; j += i) comp[j] = 1;
            }
        }
        return p;
    }();
    return table;
}
// core/lowdiscrepancy.cpp:124+: PrimeSums[i] = sum of the first i primes
inline const std::vector<int> &PrimeSums() {
    static const std::vector<int> sums = [] {
        std::vector<int> s;
        int acc = 0;
        for (int i = 0; i < PrimeTableSize; ++i) { s.push_back(acc); acc += Primes()[i]; }
        return s;
    }();
    return sums;
}

// core/lowdiscrepancy.cpp:2490-2504 + core/sampling.h:151-157 (Shuffle)
inline std::vector<uint16_t> ComputeRadicalInversePermutations(RNG &rng) {
    std::vector<uint16_t> perms;
    int permArraySize = 0;
    for (int i = 0; i < PrimeTableSize; ++i) permArraySize += Primes()[i];
    perms.resize(permArraySize);
    uint16_t *p = &perms[0];
    for (int i = 0; i < PrimeTableSize; ++i) {
        int count = Primes()[i];
        for (int j = 0; j < count; ++j) p[j] = j;
        for (int k = 0; k < count; ++k) {
            int other = k + rng.UniformUInt32(count - k);
            uint16_t t = p[k]; p[k] = p[other]; p[other] = t;
        }
        p += count;
    }
    return perms;
}

// core/lowdiscrepancy.h:67-80
inline uint32_t ReverseBits32(uint32_t n) {
    n = (n << 16) | (n >> 16);
    n = ((n & 0x00ff00ff) << 8) | ((n & 0xff00ff00) >> 8);
    n = ((n & 0x0f0f0f0f) << 4) | ((n & 0xf0f0f0f0) >> 4);
    n = ((n & 0x33333333) << 2) | ((n & 0xcccccccc) >> 2);
    n = ((n & 0x55555555) << 1) | ((n & 0xaaaaaaaa) >> 1);
    return n;
}
inline uint64_t ReverseBits64(uint64_t n) {
    uint64_t n0 = ReverseBits32((uint32_t)n);
    uint64_t n1 = ReverseBits32((uint32_t)(n >> 32));
    return (n0 << 32) | n1;
}
// core/lowdiscrepancy.cpp:389-404 (template<int base>; base is a run-time
// value here, the integer arithmetic is identical)
inline Float RadicalInverseBase(int base, uint64_t a) {
    const Float invBase = (Float)1 / (Float)base;
    uint64_t reversedDigits = 0;
    Float invBaseN = 1;
    while (a) {
        uint64_t next = a / base;
        uint64_t digit = a - next * base;
        reversedDigits = reversedDigits * base + digit;
        invBaseN *= invBase;
        a = next;
    }
    return smin(reversedDigits * invBaseN, OneMinusEpsilon);
}
// core/lowdiscrepancy.cpp:427-436
inline Float RadicalInverse(int baseIndex, uint64_t a) {
    if (baseIndex == 0) return ReverseBits64(a) * 0x1p-64;  // uint64*double -> Float
    return RadicalInverseBase(Primes()[baseIndex], a);
}
// core/lowdiscrepancy.cpp:406-424
inline Float ScrambledRadicalInverse(int baseIndex, uint64_t a, const uint16_t *perm) {
    const int base = Primes()[baseIndex];
    const Float invBase = (Float)1 / (Float)base;
    uint64_t reversedDigits = 0;
    Float invBaseN = 1;
    while (a) {
        uint64_t next = a / base;
        uint64_t digit = a - next * base;
        reversedDigits = reversedDigits * base + perm[digit];
        invBaseN *= invBase;
        a = next;
    }
    return smin(invBaseN * (reversedDigits + invBase * perm[0] / (1 - invBase)), OneMinusEpsilon);
}
// core/lowdiscrepancy.h:82-91
inline uint64_t InverseRadicalInverse(int base, uint64_t inverse, int nDigits) {
    uint64_t index = 0;
    for (int i = 0; i < nDigits; ++i) {
        uint64_t digit = inverse % base;
        inverse /= base;
        index = index * base + digit;
    }
    return index;
}

inline int64_t Mod64(int64_t a, int64_t b) { int64_t r = a - (a / b) * b; return (r < 0) ? r + b : r; }
// samplers/halton.cpp:45-62
inline void extendedGCD(uint64_t a, uint64_t b, int64_t *x, int64_t *y) {
    if (b == 0) { *x = 1; *y = 0; return; }
    int64_t d = a / b, xp, yp;
    extendedGCD(b, a % b, &xp, &yp);
    *x = yp;
    *y = xp - (d * yp);
}
inline uint64_t multiplicativeInverse(int64_t a, int64_t n) {
    int64_t x, y;
    extendedGCD(a, n, &x, &y);
    return Mod64(x, n);
}

// samplers/halton.cpp:65-131 + core/sampler.cpp:136-195 (GlobalSampler with no
// sample arrays requested: arrayStartDim == arrayEndDim == 5).
struct HaltonSampler {
    static const int kMaxResolution = 128;
    int64_t samplesPerPixel;
    int baseScales[2], baseExponents[2];
    int sampleStride;
    int multInverse[2];
    bool sampleAtPixelCenter;
    const std::vector<uint16_t> *perms;
    // per-pixel state
    int px, py;
    int64_t offsetForCurrentPixel;
    int64_t intervalSampleIndex;
    int64_t currentPixelSampleIndex;
    int dimension;

    static const std::vector<uint16_t> &Permutations() {
        static const std::vector<uint16_t> p = [] { RNG rng; return ComputeRadicalInversePermutations(rng); }();
        return p;
    }
    HaltonSampler(int spp, int sbx0, int sby0, int sbx1, int sby1, bool center = false)
        : samplesPerPixel(spp), sampleAtPixelCenter(center) {
        perms = &Permutations();
        int res[2] = {sbx1 - sbx0, sby1 - sby0};
        for (int i = 0; i < 2; ++i) {
            int base = (i == 0) ? 2 : 3;
            int scale = 1, exp = 0;
            while (scale < smin(res[i], (int)kMaxResolution)) { scale *= base; ++exp; }
            baseScales[i] = scale;
            baseExponents[i] = exp;
        }
        sampleStride = baseScales[0] * baseScales[1];
        multInverse[0] = (int)multiplicativeInverse(baseScales[1], baseScales[0]);
        multInverse[1] = (int)multiplicativeInverse(baseScales[0], baseScales[1]);
        px = py = std::numeric_limits<int>::max();
        offsetForCurrentPixel = 0; intervalSampleIndex = 0; currentPixelSampleIndex = 0; dimension = 0;
    }
    int64_t PixelOffset(int x, int y) const {
        uint64_t off = 0;
        if (sampleStride > 1) {
            int pm[2] = {(int)Mod64(x, kMaxResolution), (int)Mod64(y, kMaxResolution)};
            for (int i = 0; i < 2; ++i) {
                uint64_t dimOffset = (i == 0) ? InverseRadicalInverse(2, pm[i], baseExponents[i])
                                              : InverseRadicalInverse(3, pm[i], baseExponents[i]);
                off += dimOffset * (sampleStride / baseScales[i]) * multInverse[i];
            }
            off %= sampleStride;
        }
        return (int64_t)off;
    }
    int64_t GetIndexForSample(int64_t sampleNum) const { return offsetForCurrentPixel + sampleNum * sampleStride; }
    Float SampleDimension(int64_t index, int dim) const {
        if (sampleAtPixelCenter && (dim == 0 || dim == 1)) return 0.5f;
        if (dim == 0) return RadicalInverse(dim, index >> baseExponents[0]);
        else if (dim == 1) return RadicalInverse(dim, index / baseScales[1]);
        else return ScrambledRadicalInverse(dim, index, &(*perms)[PrimeSums()[dim]]);
    }
    void StartPixel(int x, int y) {
        px = x; py = y;
        offsetForCurrentPixel = PixelOffset(x, y);
        currentPixelSampleIndex = 0;
        dimension = 0;
        intervalSampleIndex = GetIndexForSample(0);
    }
    bool StartNextSample() {
        dimension = 0;
        intervalSampleIndex = GetIndexForSample(currentPixelSampleIndex + 1);
        return ++currentPixelSampleIndex < samplesPerPixel;
    }
    void SetSampleNumber(int64_t s) {
        dimension = 0;
        intervalSampleIndex = GetIndexForSample(s);
        currentPixelSampleIndex = s;
    }
    Float Get1D() { return SampleDimension(intervalSampleIndex, dimension++); }
    P2 Get2D() {
        P2 p(SampleDimension(intervalSampleIndex, dimension), SampleDimension(intervalSampleIndex, dimension + 1));
        dimension += 2;
        return p;
    }
};

}  // namespace orc
