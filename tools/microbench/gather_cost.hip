// gather_cost.hip — what does one 64-lane random gather cost a CU of gfx950, by width?
//
// k_hscan's first stage is one 16-byte gather per candidate and it runs at 0.7 of the texture addresser's time (profiles/r03i_sq.json).
// Would an 8-byte first stage (one {low, high} pair of the plane copy) cost half of it?  Every wave issues a long stream of
// independent random gathers of W bytes per lane (W = 4, 8, 16; addresses from a per-lane LCG, aligned to W... or to 4/8 as the
// kernel's are) into a window of S bytes: 16 KB (every line in the CU's L1), 1 MB (L2-resident, L1 misses), 64 MB (L2 misses on
// the way).  8 loads in flight per wave, 1..8 waves per SIMD.  Reported: CU cycles per wave-instruction (= shader cycles of the
// kernel x waves per CU / instructions), i.e. the texture path's cost of one gather when the CU is saturated with them.
//
// build: hipcc -O2 --offload-arch=gfx950 -o gather_cost gather_cost.hip ; run on the GPU box: ./gather_cost > gather_cost.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <string>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef uint32_t u4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2), aligned(4)));

template <int W> struct Ld;
template <> struct Ld<4>  { static __device__ __forceinline__ uint32_t ld(const uint8_t *p) { return *reinterpret_cast<const uint32_t *>(p); } };
template <> struct Ld<8>  { static __device__ __forceinline__ uint32_t ld(const uint8_t *p) { const u2 v = *reinterpret_cast<const u2 *>(p); return v.x ^ v.y; } };
template <> struct Ld<16> { static __device__ __forceinline__ uint32_t ld(const uint8_t *p) { const u4 v = *reinterpret_cast<const u4 *>(p); return v.x ^ v.y ^ v.z ^ v.w; } };

// ALIGN: address granularity in bytes (4: any word, as the packed copy's gathers; 8: the plane copy's; 64: never straddles a sector)
template <int W, int ALIGN> __global__ __launch_bounds__(256) void k_gather(const uint8_t *base, uint32_t mask, uint32_t *out, unsigned long long *clk, int iters, uint32_t seed)
{
    uint32_t x = seed * (blockIdx.x * 256 + threadIdx.x + 1) * 2654435761u + 12345u;
    uint32_t acc = 0;
    // each block works in its own window of the buffer where the buffer is larger than the window (so that L1 / L2 see `mask + 1` bytes per CU or per chip as asked)
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        uint32_t a[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { x = x * 1664525u + 1013904223u; a[k] = (x >> 4) & mask & ~(uint32_t)(ALIGN - 1); }
#pragma unroll
        for (int k = 0; k < 8; k++) acc ^= Ld<W>::ld(base + a[k]);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) { clk[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0; clk[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = r1 - r0; }
}

template <int W, int ALIGN> static int run(int n_cu, const uint8_t *d_buf, uint32_t window, uint32_t *d_out, unsigned long long *d_clk, std::string &json, bool first)
{
    char buf[512];
    snprintf(buf, sizeof buf, "%s  {\"bytes_per_lane\": %d, \"align\": %d, \"window_bytes\": %u, \"by_waves_per_simd\": {", first ? "" : ",\n", W, ALIGN, window);
    json += buf;
    const int wps[] = {1, 2, 4, 6, 8};
    const int iters = 2048;
    for (int wi = 0; wi < 5; wi++) {
        const int w = wps[wi], blocks = n_cu * w;
        hipLaunchKernelGGL((k_gather<W, ALIGN>), dim3(blocks), dim3(256), 0, 0, d_buf, window - 1, d_out, d_clk, 64, 777u);
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_gather<W, ALIGN>), dim3(blocks), dim3(256), 0, 0, d_buf, window - 1, d_out, d_clk, iters, 777u);
        CHK(hipEventRecord(e1, 0));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> clk((size_t)blocks * 8);
        CHK(hipMemcpy(clk.data(), d_clk, clk.size() * 8, hipMemcpyDeviceToHost));
        const double n_instr = (double)blocks * 4 * iters * 8;          // wave-instructions of the whole launch
        const double per_cu_s = n_instr / n_cu / (ms * 1e-3);           // wave-instructions per CU and second
        std::vector<double> ghz;
        for (int i = 0; i < blocks * 4; i++) ghz.push_back((double)clk[2 * i] / ((double)clk[2 * i + 1] * 10.0));
        std::sort(ghz.begin(), ghz.end());
        snprintf(buf, sizeof buf, "%s\"%d\": {\"kernel_ms\": %.3f, \"chip_G_lane_loads_per_s\": %.1f, \"ns_per_wave_instr_per_cu\": %.2f, \"cu_cycles_per_wave_instr_at_2p4GHz\": %.1f}",
                 wi ? ", " : "", w, ms, n_instr * 64 / (ms * 1e-3) / 1e9, 1e9 / per_cu_s, 2.4e9 / per_cu_s);
        json += buf;
        CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
    }
    json += "}}";
    return 0;
}

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    uint8_t *d_buf; uint32_t *d_out; unsigned long long *d_clk;
    const size_t bytes = (size_t)64 << 20;
    CHK(hipMalloc((void **)&d_buf, bytes + 256));
    CHK(hipMemset(d_buf, 1, bytes + 256));
    CHK(hipMalloc((void **)&d_out, (size_t)n_cu * 8 * 256 * 4));
    CHK(hipMalloc((void **)&d_clk, (size_t)n_cu * 8 * 8 * 8));
    std::string json = "{\"device\": \"" + std::string(prop.gcnArchName) + "\", \"cus\": " + std::to_string(n_cu) + ", \"gathers\": [\n";
    bool first = true;
    for (uint32_t window : {16u << 10, 1u << 20, 64u << 20}) {
        if (run<4, 4>(n_cu, d_buf, window, d_out, d_clk, json, first)) return 1;
        first = false;
        if (run<8, 4>(n_cu, d_buf, window, d_out, d_clk, json, false)) return 1;
        if (run<8, 8>(n_cu, d_buf, window, d_out, d_clk, json, false)) return 1;
        if (run<16, 4>(n_cu, d_buf, window, d_out, d_clk, json, false)) return 1;
        if (run<16, 8>(n_cu, d_buf, window, d_out, d_clk, json, false)) return 1;
        if (run<16, 16>(n_cu, d_buf, window, d_out, d_clk, json, false)) return 1;
    }
    json += "\n]}\n";
    fputs(json.c_str(), stdout);
    return 0;
}
