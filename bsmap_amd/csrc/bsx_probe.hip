// bsx_probe.hip — measured memory ceilings of the device the library runs on ("report both peak and a
// measured ceiling"): a streaming read, a streaming copy and the access pattern the scan kernels are made of — 16-byte
// loads at random 4-byte-aligned addresses.  bench.py puts the three rates into roofline.peak_measured; nothing on the
// alignment path calls this.
#include "bsx_internal.h"

namespace {

typedef unsigned long long u64;
struct __attribute__((packed, aligned(4))) U4u { uint32_t a, b, c, d; };
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_probe_fill(uint4 *p, u64 n16)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n16; i += (u64)gridDim.x * 256) {
        const uint32_t x = (uint32_t)i * 2654435761u;
        p[i] = make_uint4(x, x ^ 0x9E3779B9u, x + 77u, ~x);
    }
}

__global__ __launch_bounds__(256) void k_probe_read(const v4u *__restrict__ p, u64 n16, uint32_t *sink)
{
    uint32_t s = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n16; i += (u64)gridDim.x * 256) {
        const v4u v = __builtin_nontemporal_load(p + i);
        s += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (s == 0x12345678u) *sink = s;  // never true in practice; keeps the loads alive
}

__global__ __launch_bounds__(256) void k_probe_copy(const v4u *__restrict__ src, v4u *__restrict__ dst, u64 n16)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n16; i += (u64)gridDim.x * 256) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

// every lane: `per_lane` loads of 16 bytes at pseudo-random 4-byte-aligned word offsets inside [0, n_words - 4)
__global__ __launch_bounds__(256) void k_probe_gather(const uint32_t *__restrict__ p, uint32_t n_words, uint32_t per_lane, uint32_t *sink)
{
    uint32_t s = 0, x = (blockIdx.x * 256u + threadIdx.x) * 747796405u + 2891336453u;
    for (uint32_t i = 0; i < per_lane; i += 4) {
        uint32_t o[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { x = x * 1664525u + 1013904223u; o[u] = (uint32_t)(((u64)(x ^ (x >> 15)) * (u64)(n_words - 4)) >> 32); }
        U4u v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const U4u *>(p + o[u]);
#pragma unroll
        for (int u = 0; u < 4; u++) s += v[u].a ^ v[u].b ^ v[u].c ^ v[u].d;
    }
    if (s == 0x12345678u) *sink = s;
}

}  // namespace

// rates in GB/s (1e9 bytes): stream read of `bytes`, copy (read + write counted), random 16-byte gathers over a window of
// `gather_window_bytes` (useful bytes = 16 per load; the memory system moves a 64-byte or 128-byte sector for each)
extern "C" int bsx_probe_memory(int device, uint64_t bytes, uint64_t gather_window_bytes, double *read_GBps, double *copy_GBps, double *gather16_GBps,
                                double *gather16_Gloads_per_s)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) { g_bsx_err = "no such device"; return BSX_ERR_NODEVICE; }
    HIP_TRY(hipSetDevice(device));
    if (bytes < (1u << 20) || gather_window_bytes < 4096 || gather_window_bytes > bytes || gather_window_bytes > (1ull << 33)) return BSX_ERR_ARG;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    const u64 n16 = bytes / 16;
    uint4 *a = nullptr, *b = nullptr;
    uint32_t *sink = nullptr;
    HIP_TRY(hipMalloc((void **)&a, n16 * 16));
    if (hipMalloc((void **)&b, n16 * 16) != hipSuccess) { (void)hipFree(a); return BSX_ERR_NOMEM; }
    if (hipMalloc((void **)&sink, 256) != hipSuccess) { (void)hipFree(a); (void)hipFree(b); return BSX_ERR_NOMEM; }
    const int grid = prop.multiProcessorCount * 8;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_probe_fill, dim3(grid), dim3(256), 0, 0, a, n16);
    hipLaunchKernelGGL(k_probe_fill, dim3(grid), dim3(256), 0, 0, b, n16);
    auto timed = [&](auto launch, int reps, float &best) -> int {
        best = 1e30f;
        for (int r = 0; r < reps + 1; r++) {
            HIP_TRY(hipEventRecord(e0, 0));
            launch();
            HIP_TRY(hipEventRecord(e1, 0));
            HIP_TRY(hipEventSynchronize(e1));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0 && ms < best) best = ms;  // first repetition is the warm-up
        }
        return BSX_OK;
    };
    float ms = 0;
    int rc = timed([&] { hipLaunchKernelGGL(k_probe_read, dim3(grid), dim3(256), 0, 0, (const v4u *)a, n16, sink); }, 5, ms);
    if (rc == BSX_OK && read_GBps) *read_GBps = (double)bytes / (ms * 1e-3) / 1e9;
    if (rc == BSX_OK) rc = timed([&] { hipLaunchKernelGGL(k_probe_copy, dim3(grid), dim3(256), 0, 0, (const v4u *)a, (v4u *)b, n16); }, 5, ms);
    if (rc == BSX_OK && copy_GBps) *copy_GBps = 2.0 * (double)bytes / (ms * 1e-3) / 1e9;
    const uint32_t per_lane = 256, n_words = (uint32_t)(gather_window_bytes / 4);
    const int ggrid = prop.multiProcessorCount * 16;
    if (rc == BSX_OK) rc = timed([&] { hipLaunchKernelGGL(k_probe_gather, dim3(ggrid), dim3(256), 0, 0, (const uint32_t *)a, n_words, per_lane, sink); }, 3, ms);
    const double loads = (double)ggrid * 256.0 * per_lane;
    if (rc == BSX_OK && gather16_GBps) *gather16_GBps = loads * 16.0 / (ms * 1e-3) / 1e9;
    if (rc == BSX_OK && gather16_Gloads_per_s) *gather16_Gloads_per_s = loads / (ms * 1e-3) / 1e9;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(sink);
    return rc;
}
