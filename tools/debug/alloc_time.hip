// How long does a fresh process wait for 100 GB of device memory?  hipMalloc vs hipMallocAsync (stream-ordered pool).
// hipcc --offload-arch=gfx950 -O2 tools/debug/alloc_time.hip -o /tmp/alloc_time && /tmp/alloc_time [gb] [mode 0|1]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void touch(char *p, size_t n, size_t stride) { size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * stride; if (i < n) p[i] = 1; }
int main(int argc, char **argv) {
    const size_t gb = argc > 1 ? (size_t)atoi(argv[1]) : 100;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    const size_t n = gb << 30;
    (void)hipFree(nullptr);
    auto t0 = std::chrono::high_resolution_clock::now();
    char *p = nullptr;
    hipError_t e;
    if (mode == 0) e = hipMalloc((void **)&p, n);
    else { e = hipMallocAsync((void **)&p, n, nullptr); if (e == hipSuccess) e = hipStreamSynchronize(nullptr); }
    auto t1 = std::chrono::high_resolution_clock::now();
    if (e != hipSuccess) { std::printf("alloc failed: %s\n", hipGetErrorString(e)); return 1; }
    hipLaunchKernelGGL(touch, dim3((unsigned)((n / 4096 + 255) / 256)), dim3(256), 0, nullptr, p, n, (size_t)4096);
    e = hipDeviceSynchronize();
    auto t2 = std::chrono::high_resolution_clock::now();
    std::printf("mode %d (%s): %zu GB: alloc %.3f s, first touch of every page %.3f s (%s)\n", mode, mode ? "hipMallocAsync" : "hipMalloc", gb,
                std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count(), hipGetErrorString(e));
    return 0;
}
