// hprt host side — tables the device Halton sampler reads: the first 1000 primes,
// their prefix sums and the digit permutations that the pbrt-v3 fork's HaltonSampler
// derives from a default-seeded PCG32 (samplers/halton.cpp:69-72,
// core/lowdiscrepancy.cpp:2490-2504, core/sampling.h:151-157, core/rng.h:61-144).
#pragma once
#include <cstdint>
#include <vector>

namespace hprt {

const int kPrimeTableSize = 1000;
const std::vector<int> &PrimeTable();                  // Primes[], lowdiscrepancy.cpp:40
const std::vector<int> &PrimeSumTable();               // PrimeSums[], lowdiscrepancy.cpp:124
const std::vector<uint16_t> &HaltonPermutations();     // radicalInversePermutations

struct HaltonLayout {          // HaltonSampler ctor, samplers/halton.cpp:65-98
    int baseScales[2], baseExponents[2];
    int sampleStride;
    int multInverse[2];
};
HaltonLayout MakeHaltonLayout(int resX, int resY);
// GetIndexForSample's per-pixel offset (samplers/halton.cpp:101-120)
int64_t HaltonPixelOffset(const HaltonLayout &h, int px, int py);

}  // namespace hprt
