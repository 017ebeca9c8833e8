// launch_cost.hip — what a device-driven heavy pipeline would pay for worst-case-sized launches (round-3 design evidence):
//   * dispatch of a grid whose blocks exit at once (scan grid sized for the whole task pool, a few tasks present)
//   * a chain of dependent small kernels in one stream (launch-to-launch latency)
//   * rocPRIM radix sort of (key, id) pairs at the pool size against the typical task count
// build: hipcc -O2 --offload-arch=gfx950 -o launch_cost launch_cost.hip ; prints one JSON object
#include <hip/hip_runtime.h>
#include <cstring>
#include <functional>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <cstdint>

__global__ void k_exit(const uint32_t *n, uint32_t *sink) { if (blockIdx.x * 2u + (threadIdx.x >> 6) < *n) atomicAdd(sink, 1u); }
__global__ void k_tiny(uint32_t *p) { if (threadIdx.x == 0) p[0] += 1; }

static float time_it(hipStream_t s, int reps, const std::function<void()> &f)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(b, s); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / reps;  // microseconds per call
}

int main()
{
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    uint32_t *d_n, *d_sink; hipMalloc(&d_n, 4); hipMalloc(&d_sink, 4); hipMemset(d_n, 0, 4); hipMemset(d_sink, 0, 4);
    printf("{\"empty_grid_us\": {");
    const uint32_t grids[] = {1024, 8192, 32768, 131072, 262144, 1048576};
    for (int i = 0; i < 6; i++) {
        const uint32_t g = grids[i];
        float us = time_it(s, 50, [&] { hipLaunchKernelGGL(k_exit, dim3(g), dim3(128), 0, s, d_n, d_sink); });
        printf("%s\"%u\": %.2f", i ? ", " : "", g, us);
    }
    printf("}, ");
    float chain = time_it(s, 200, [&] { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, d_sink); });
    printf("\"dependent_tiny_kernel_us\": %.2f, ", chain);
    // graph of 64 tiny kernels
    {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < 64; i++) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, d_sink);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        float us = time_it(s, 20, [&] { hipGraphLaunch(ge, s); });
        printf("\"graph_64_tiny_kernels_us\": %.2f, ", us);
    }
    const uint32_t cap = 262144;
    uint32_t *k0, *k1, *v0, *v1; hipMalloc(&k0, cap * 4); hipMalloc(&k1, cap * 4); hipMalloc(&v0, cap * 4); hipMalloc(&v1, cap * 4);
    std::vector<uint32_t> h(cap);
    for (uint32_t i = 0; i < cap; i++) h[i] = (i * 2654435761u) % 1475952693u;
    hipMemcpy(k0, h.data(), cap * 4, hipMemcpyHostToDevice); hipMemcpy(v0, h.data(), cap * 4, hipMemcpyHostToDevice);
    size_t need = 0; rocprim::radix_sort_pairs(nullptr, need, k0, k1, v0, v1, (size_t)cap, 0u, 32u, s);
    void *tmp; hipMalloc(&tmp, need * 2);
    printf("\"rocprim_sort_pairs_us\": {");
    const uint32_t ns[] = {64, 1024, 8192, 65536, 262144};
    for (int i = 0; i < 5; i++) {
        const uint32_t n = ns[i];
        size_t nd = need * 2;
        float us = time_it(s, 50, [&] { rocprim::radix_sort_pairs(tmp, nd, k0, k1, v0, v1, (size_t)n, 0u, 32u, s); });
        printf("%s\"%u\": %.2f", i ? ", " : "", n, us);
    }
    printf("}}\n");
    return 0;
}
