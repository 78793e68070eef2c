// Micro-benchmark: how fast can a CU gather small records through its vector L1?
// Answers the question DESIGN.md section 7 asks of k_trace: is the per-step fetch of a 64-byte child-pair record by every lane
// bounded by L1 requests (one per lane and b128), by L1 bytes, or by latency?
//   A  every lane loads its own 64-byte record with four b128 loads        (what k_trace does)
//   B  the four lanes of a quad load the four 16-byte pieces of ONE record, four rounds   (same bytes, contiguous within a quad)
//   C  every lane loads a 32-byte record (two b128)
//   D  every lane loads a 16-byte record (one b128)
// Records are picked by a per-lane LCG (independent loads) or by the previous record's contents (dependent, like a traversal).
// build: hipcc --offload-arch=gfx950 -O3 -o l1_gather_bench tools/debug/l1_gather_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t quad_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ uint32_t quad_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true); }   // quad_perm [2,3,0,1]
// 2x2 exchange of (x0, x1) with the partner lane: the lane whose bit is clear keeps x0 and receives the partner's x0 in x1, and vice versa
template <int STAGE> __device__ __forceinline__ void xchg(uint32_t &x0, uint32_t &x1, bool bit) {
    const uint32_t send = bit ? x0 : x1;
    const uint32_t recv = STAGE == 1 ? quad_xor1(send) : quad_xor2(send);
    x0 = bit ? recv : x0; x1 = bit ? x1 : recv;
}
__device__ __forceinline__ void xchg4(uint4 &a, uint4 &b, bool bit, int stage) {
    if (stage == 1) { xchg<1>(a.x, b.x, bit); xchg<1>(a.y, b.y, bit); xchg<1>(a.z, b.z, bit); xchg<1>(a.w, b.w, bit); }
    else { xchg<2>(a.x, b.x, bit); xchg<2>(a.y, b.y, bit); xchg<2>(a.z, b.z, bit); xchg<2>(a.w, b.w, bit); }
}
// TREE: records are picked like BVH nodes (a level uniformly in 0..log2(n)-1, then a node of that level): the top of the tree is hot
template <int VAR, bool DEP, int PAD, bool TREE>
__global__ __launch_bounds__(256, 6) void k(const uint4 *__restrict__ rec, uint32_t mask, int iters, uint32_t *out) {
    __shared__ uint32_t pad[6144];      // 24 KB: the occupancy k_trace runs at (6 blocks of 256 per CU)
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    if (threadIdx.x == 999) pad[0] = 1;
    for (int it = 0; it < iters; ++it) {
        idx = idx * 1664525u + 1013904223u;
        uint32_t r = (idx >> 8) & mask;
        if (TREE) { const uint32_t levels = 32u - __clz(mask); const uint32_t l = (idx >> 3) % levels; r = (1u << l) | (r & ((1u << l) - 1u)); r &= mask; }
        float padv = __uint_as_float((idx & 0x7fffffu) | 0x3f800000u);
        if (VAR == 0) {
            const uint4 a = rec[4 * r], b = rec[4 * r + 1], c = rec[4 * r + 2], d = rec[4 * r + 3];
            acc += a.x ^ b.y ^ c.z ^ d.w;
            if (DEP) idx ^= a.x;
        } else if (VAR == 1) {
            const uint32_t s = threadIdx.x & 3u;
            uint32_t t = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t rq = __shfl(r, (threadIdx.x & ~3u) | q, 64);
                const uint4 a = rec[4 * rq + s];
                t ^= a.x ^ a.y ^ a.z ^ a.w;
                if (DEP && q == (int)s) idx ^= a.x;
            }
            acc += t;
        } else if (VAR == 4) {
            const uint32_t s = threadIdx.x & 3u;
            uint4 R[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t rq = __shfl(r, (threadIdx.x & ~3u) | q, 64);
                R[q] = rec[4 * rq + s];
            }
            const bool b0 = (s & 1u) != 0u, b1 = (s & 2u) != 0u;
            xchg4(R[0], R[1], b0, 1); xchg4(R[2], R[3], b0, 1);
            xchg4(R[0], R[2], b1, 2); xchg4(R[1], R[3], b1, 2);
            acc += R[0].x ^ R[1].y ^ R[2].z ^ R[3].w;
            if (DEP) idx ^= R[0].x;
            padv += __uint_as_float((R[1].x & 0x7fffffu) | 0x3f800000u);
        } else if (VAR == 2) {
            const uint4 a = rec[4 * r], b = rec[4 * r + 1];
            acc += a.x ^ b.y;
            if (DEP) idx ^= a.x;
        } else {
            const uint4 a = rec[4 * r];
            acc += a.x;
            if (DEP) idx ^= a.x;
        }
#pragma unroll
        for (int p = 0; p < PAD; ++p) padv = padv * 1.0000001f + 0.5f;
        acc += __float_as_uint(padv) & 1u;
        if (DEP) idx ^= __float_as_uint(padv) & 1u;
    }
    if (acc == 0x12345678u) out[0] = acc + pad[threadIdx.x];
}

template <int VAR, bool DEP, int PAD = 0, bool TREE = false>
static void run(const char *name, const uint4 *rec, uint32_t nRec, uint32_t *out, int nCU, double mhz) {
    const int iters = 2000, blocks = nCU * 6 * 4;
    k<VAR, DEP, PAD, TREE><<<blocks, 256>>>(rec, nRec - 1, 100, out);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    k<VAR, DEP, PAD, TREE><<<blocks, 256>>>(rec, nRec - 1, iters, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double recs = (double)blocks * 256 * iters;
    const double perClkCU = recs / (ms * 1e-3) / (mhz * 1e6) / nCU;
    printf("  %-34s %-4s pad %3d %s %8.2f ms  %8.1f Grecords/s  %6.3f records/clk/CU\n", name, DEP ? "dep" : "ind", PAD, TREE ? "tree" : "flat", ms, recs / ms * 1e-6, perClkCU);
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int nCU = p.multiProcessorCount; const double mhz = p.clockRate / 1000.0;
    printf("%s: %d CUs, %.0f MHz\n", p.name, nCU, mhz);
    uint32_t *out; CK(hipMalloc(&out, 4));
    for (uint32_t nRec : {256u, 16384u, 1u << 18, 1u << 20, 1u << 23}) {      // 16 KB (L1), 1 MB (L2), 64 MB (MALL), 512 MB (HBM)
        std::vector<uint32_t> h((size_t)nRec * 16);
        uint32_t s = 12345u;
        for (auto &v : h) { s = s * 1103515245u + 12345u; v = s; }
        uint4 *rec; CK(hipMalloc(&rec, h.size() * 4)); CK(hipMemcpy(rec, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        printf("%u records of 64 B (%.2f MB)\n", nRec, nRec * 64.0 / 1048576);
        run<0, false>("A per-lane 64 B (4 x b128)", rec, nRec, out, nCU, mhz);
        run<1, false>("B quad-cooperative 64 B (4 rounds)", rec, nRec, out, nCU, mhz);
        run<2, false>("C per-lane 32 B (2 x b128)", rec, nRec, out, nCU, mhz);
        run<3, false>("D per-lane 16 B (1 x b128)", rec, nRec, out, nCU, mhz);
        run<0, true>("A per-lane 64 B (4 x b128)", rec, nRec, out, nCU, mhz);
        run<1, true>("B quad-cooperative 64 B (4 rounds)", rec, nRec, out, nCU, mhz);
        run<2, true>("C per-lane 32 B (2 x b128)", rec, nRec, out, nCU, mhz);
        run<3, true>("D per-lane 16 B (1 x b128)", rec, nRec, out, nCU, mhz);
        run<4, false>("E quad-coop + quad transpose", rec, nRec, out, nCU, mhz);
        run<4, true>("E quad-coop + quad transpose", rec, nRec, out, nCU, mhz);
        if (nRec == (1u << 20) / 4) {
            printf(" tree-like picks, with dependent ALU work per record:\n");
            run<0, true, 0, true>("A per-lane 64 B", rec, nRec, out, nCU, mhz);
            run<4, true, 0, true>("E quad-coop + transpose", rec, nRec, out, nCU, mhz);
            run<0, true, 50, true>("A per-lane 64 B", rec, nRec, out, nCU, mhz);
            run<4, true, 50, true>("E quad-coop + transpose", rec, nRec, out, nCU, mhz);
            run<0, true, 100, true>("A per-lane 64 B", rec, nRec, out, nCU, mhz);
            run<4, true, 100, true>("E quad-coop + transpose", rec, nRec, out, nCU, mhz);
            run<0, true, 200, true>("A per-lane 64 B", rec, nRec, out, nCU, mhz);
            run<4, true, 200, true>("E quad-coop + transpose", rec, nRec, out, nCU, mhz);
        }
        CK(hipFree(rec));
    }
    return 0;
}
