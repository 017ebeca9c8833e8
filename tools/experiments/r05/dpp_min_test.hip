// GPU box: wave_min_u32 (DPP) against a plain reduction, on random and adversarial lanes.  hipcc --offload-arch=gfx950 -O3 dpp_min_test.hip -o dpp_min_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
__device__ __forceinline__ uint32_t rl(uint32_t x, int l) { return __builtin_amdgcn_readlane(x, l); }
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#define BSX_DPP_MIN(ctrl, rows) v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, ctrl, rows, 0xf, false))
    BSX_DPP_MIN(0x111, 0xf); BSX_DPP_MIN(0x112, 0xf); BSX_DPP_MIN(0x114, 0xf); BSX_DPP_MIN(0x118, 0xf);
    BSX_DPP_MIN(0x142, 0xa);
    BSX_DPP_MIN(0x143, 0xc);
#undef BSX_DPP_MIN
    return rl(v, 63);
}
__global__ void k(const uint32_t *in, uint32_t *out, int n)
{
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const uint32_t v = in[i * 64 + threadIdx.x];
        const uint32_t m = wave_min_u32(v);
        if (threadIdx.x == 0) out[i] = m;
    }
}
int main()
{
    const int n = 4096;
    uint32_t *h = (uint32_t *)malloc(n * 64 * 4), *d, *o, *ho = (uint32_t *)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; i++)
        for (int l = 0; l < 64; l++) {
            uint32_t v = (uint32_t)rand() * 2654435761u;
            if (i % 4 == 1) v = (l == i % 64) ? 5u : 0xffffffffu;      // one competing lane
            if (i % 4 == 2) v = 0xffffffffu;                            // none
            if (i % 4 == 3) v = (l >= (i % 16)) ? 0xffffffffu : (uint32_t)(1000 - l + (i >> 4));
            h[i * 64 + l] = v;
        }
    hipMalloc(&d, n * 64 * 4); hipMalloc(&o, n * 4);
    hipMemcpy(d, h, n * 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(256), dim3(64), 0, 0, d, o, n);
    hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) {
        uint32_t m = 0xffffffffu;
        for (int l = 0; l < 64; l++) m = h[i * 64 + l] < m ? h[i * 64 + l] : m;
        if (m != ho[i]) { if (bad < 5) printf("row %d: want %u got %u\n", i, m, ho[i]); bad++; }
    }
    printf("%d of %d wrong\n", bad, n);
    return bad != 0;
}
