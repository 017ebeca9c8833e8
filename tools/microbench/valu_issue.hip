// valu_issue.hip — how many cycles does one SIMD of gfx950 need per wave64 VALU instruction of the kind k_hscan issues?
//
// Settles the 2- vs 4-cycle question of VERDICT r01 (weak #5): every wave runs a long, fully unrolled stream of
// register-only instructions (8 independent chains, so no dependency stall), with 1 / 2 / 4 / 6 / 8 waves resident per
// SIMD (grid = CUs x waves-per-SIMD blocks of 256 threads; the kernel uses < 32 VGPRs, so all of them are co-resident).
// Reported per mix: cycles per wave-instruction seen by one wave (s_memtime delta / instructions) and the SIMD's
// issue cost = that / waves-per-SIMD, i.e. shader cycles one SIMD spends per wave64 instruction when its waves interleave.
// Mixes: the exact k_hscan head word (alignbit, bitop3, lshl, bitop3, bcnt), each opcode alone, v_fma_f32 for reference.
//
// build: hipcc -O2 --offload-arch=gfx950 -o valu_issue valu_issue.hip ; run on the GPU box: ./valu_issue > valu_issue.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <string>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// one "slot" = one instruction on chain c (registers a[c], b[c]); BODY(c) must be exactly N_PER_SLOT instructions
#define REP8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)

#define MIX_XOR(c)   asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
#define MIX_BCNT(c)  asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[c]) : "v"(b[c]));
#define MIX_ALIGN(c) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b[c]), "v"(sh));
#define MIX_BITOP(c) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x60" : "+v"(a[c]) : "v"(b[c]), "v"(sh));
#define MIX_BITOPS(c) asm volatile("v_bitop3_b32 %0, %2, %0, %1 bitop3:0x60" : "+v"(a[c]) : "v"(b[c]), "s"(sc));
#define MIX_LSHL(c)  asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[c]));
#define MIX_FMA(c)   asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(b[c]));
#define MIX_MUL24(c) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
#define MIX_MAD24(c) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[c]) : "v"(b[c]));
#define MIX_ADD3(c)  asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(b[c]));
#define MIX_CMP(c)   asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(a[c]), "v"(b[c]) : "vcc");
#define MIX_CMPADDC(c) asm volatile("v_cmp_gt_u32 vcc, %1, %0\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(a[c]) : "v"(b[c]) : "vcc");
#define MIX_CNDMASK(c) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(b[c]) : "vcc");
// the head word of k_hscan: funnel shift, (read ^ ref) & tmask, << 1, (t | y) & 0xAAAAAAAA, popcount (5 instructions)
#define MIX_HEAD(c)  asm volatile("v_alignbit_b32 %0, %0, %1, %2\n v_bitop3_b32 %0, %3, %0, %1 bitop3:0x60\n v_lshlrev_b32 %1, 1, %0\n" \
                                  "v_bitop3_b32 %0, %1, %4, %0 bitop3:0xc8\n v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[c]), "+v"(b[c]) : "v"(sh), "s"(sc), "s"(sd));
// the same word as k_hscan issues it since round 3: every operand in a VGPR, the shift as an add
#define MIX_HEADV(c) asm volatile("v_alignbit_b32 %0, %0, %1, %2\n v_bitop3_b32 %0, %3, %0, %1 bitop3:0x60\n v_add_u32 %1, %0, %0\n" \
                                  "v_bitop3_b32 %0, %1, %4, %0 bitop3:0xc8\n v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[c]), "+v"(b[c]) : "v"(sh), "v"(sc), "v"(sd));
#define MIX_ADD(c)   asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
// The inner word of k_hscan_same as the shipped code object issues it (hs_eval_read<5, false, PLAIN, FULL>: one read of 129-160 nt without N
// against 64 candidates): per 32-nt word two v_bitop3 (three for the last, masked one) and a v_bcnt that accumulates; one compare into an
// SGPR pair for the survivors; plus the group's ten v_alignbit per candidate shared by 9.6 reads on average (one here).
//   counted (bsx_batch_set_work_counters(1)): + the reference's two early-out classes — 2 v_and, 2 v_bcnt, 2 compares.
#define HS_WORD2(c)  "v_bitop3_b32 %1, %0, %3, %2 bitop3:0x34\n v_bitop3_b32 %1, %1, %0, %2 bitop3:0xf6\n v_bcnt_u32_b32 %0, %1, %0\n"
#define HS_WORD3(c)  "v_bitop3_b32 %1, %0, %3, %2 bitop3:0x28\n v_bitop3_b32 %0, %0, %2, %3 bitop3:0x34\n v_bitop3_b32 %1, %1, %0, %2 bitop3:0xf8\n v_bcnt_u32_b32 %0, %1, %0\n"
#define MIX_HSAME(c) asm volatile("v_alignbit_b32 %0, %0, %1, %2\n" HS_WORD2(c) HS_WORD2(c) HS_WORD2(c) HS_WORD2(c) HS_WORD3(c) \
                                  "v_cmp_ge_u32_e64 s[10:11], %3, %0" : "+v"(a[c]), "+v"(b[c]) : "v"(sh), "v"(sc) : "s10", "s11");
#define MIX_HSAMEC(c) asm volatile("v_alignbit_b32 %0, %0, %1, %2\n" HS_WORD2(c) "v_and_b32 %1, %1, %3\n v_bcnt_u32_b32 %1, %1, 0\n v_cmp_lt_u32_e64 s[12:13], %3, %1\n" \
                                   HS_WORD2(c) "v_and_b32 %1, %1, %3\n v_bcnt_u32_b32 %1, %1, %0\n v_cmp_ge_u32_e64 s[14:15], %3, %1\n" HS_WORD2(c) HS_WORD2(c) HS_WORD3(c) \
                                   "v_cmp_ge_u32_e64 s[10:11], %3, %0" : "+v"(a[c]), "+v"(b[c]) : "v"(sh), "v"(sc) : "s10", "s11", "s12", "s13", "s14", "s15");

template <int MIX> struct Info;
#define DEF(ID, NAME, BODY, N) \
    template <> struct Info<ID> { static constexpr int n = N; static const char *name() { return NAME; } \
        static __device__ __forceinline__ void run(uint32_t (&a)[8], uint32_t (&b)[8], uint32_t sh, uint32_t sc, uint32_t sd) { REP8(BODY) REP8(BODY) REP8(BODY) REP8(BODY) } };
DEF(0, "v_xor_b32", MIX_XOR, 1)
DEF(1, "v_bcnt_u32_b32", MIX_BCNT, 1)
DEF(2, "v_alignbit_b32", MIX_ALIGN, 1)
DEF(3, "v_bitop3_b32 (vvv)", MIX_BITOP, 1)
DEF(4, "v_bitop3_b32 (svv)", MIX_BITOPS, 1)
DEF(5, "v_lshlrev_b32", MIX_LSHL, 1)
DEF(6, "v_fma_f32", MIX_FMA, 1)
DEF(7, "v_mul_u32_u24", MIX_MUL24, 1)
DEF(8, "v_mad_u32_u24", MIX_MAD24, 1)
DEF(9, "v_add3_u32", MIX_ADD3, 1)
DEF(10, "v_cmp_gt_u32 (vcc)", MIX_CMP, 1)
DEF(11, "v_cmp + v_addc_co", MIX_CMPADDC, 2)
DEF(12, "v_cndmask_b32", MIX_CNDMASK, 1)
DEF(13, "k_hscan head word (alignbit,bitop3,lshl,bitop3,bcnt)", MIX_HEAD, 5)
DEF(14, "v_add_u32", MIX_ADD, 1)
DEF(15, "k_hscan head word, VGPR operands (alignbit,bitop3 vvv,add,bitop3 vvv,bcnt)", MIX_HEADV, 5)
DEF(16, "k_hscan_same inner word (alignbit, 11 bitop3 vvv, 5 bcnt, 1 cmp -> sgpr)", MIX_HSAME, 18)
DEF(17, "k_hscan_same inner word with work counters (alignbit, 11 bitop3 vvv, 7 bcnt, 2 and, 3 cmp -> sgpr)", MIX_HSAMEC, 24)
constexpr int N_MIX = 18;

template <int MIX> __global__ __launch_bounds__(256) void k_issue(uint32_t *out, unsigned long long *clk, int iters, uint32_t seed)
{
    uint32_t a[8], b[8];
    for (int c = 0; c < 8; c++) { a[c] = seed * (threadIdx.x + 1) + c; b[c] = (seed ^ 0x9E3779B9u) * (c + 3) + threadIdx.x; }
    const uint32_t sh = (threadIdx.x * 2) & 31, sc = seed * 77u, sd = 0xAAAAAAAAu;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) Info<MIX>::run(a, b, sh, sc, sd);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t s = 0;
    for (int c = 0; c < 8; c++) s += a[c] ^ b[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { clk[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0; clk[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = r1 - r0; }
}

template <int MIX> static int run_mix(int n_cu, uint32_t *d_out, unsigned long long *d_clk, std::string &json)
{
    const int iters = 4096;
    const double instr_per_wave = (double)iters * 32 * Info<MIX>::n;
    char buf[512];
    snprintf(buf, sizeof buf, "  {\"mix\": \"%s\", \"instructions_per_wave\": %.0f, \"by_waves_per_simd\": {", Info<MIX>::name(), instr_per_wave);
    json += buf;
    const int wps[] = {1, 2, 4, 6, 8};
    for (int wi = 0; wi < 5; wi++) {
        const int w = wps[wi], blocks = n_cu * w;
        hipLaunchKernelGGL(k_issue<MIX>, dim3(blocks), dim3(256), 0, 0, d_out, d_clk, 64, 12345u);  // warm-up
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_issue<MIX>, dim3(blocks), dim3(256), 0, 0, d_out, d_clk, iters, 12345u);
        CHK(hipEventRecord(e1, 0));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> clk((size_t)blocks * 8);
        CHK(hipMemcpy(clk.data(), d_clk, clk.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> cyc, ghz;
        for (int i = 0; i < blocks * 4; i++) { cyc.push_back((double)clk[2 * i]); ghz.push_back((double)clk[2 * i] / ((double)clk[2 * i + 1] * 10.0)); }  // s_memrealtime ticks at 100 MHz
        std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
        const double med = cyc[cyc.size() / 2], per_wave = med / instr_per_wave, per_simd = per_wave / w;
        const double chip_rate = (double)blocks * 4 * instr_per_wave / (ms * 1e-3);
        snprintf(buf, sizeof buf, "%s\"%d\": {\"cycles_per_instr_one_wave\": %.3f, \"simd_cycles_per_wave_instr\": %.3f, \"kernel_ms\": %.3f, \"chip_G_wave_instr_per_s\": %.1f, \"shader_GHz\": %.3f}",
                 wi ? ", " : "", w, per_wave, per_simd, ms, chip_rate / 1e9, ghz[ghz.size() / 2]);
        json += buf;
        CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
    }
    json += "}}";
    return 0;
}

template <int MIX> static int run_all(int n_cu, uint32_t *d_out, unsigned long long *d_clk, std::string &json)
{
    if (run_mix<MIX>(n_cu, d_out, d_clk, json)) return 1;
    if constexpr (MIX + 1 < N_MIX) { json += ",\n"; return run_all<MIX + 1>(n_cu, d_out, d_clk, json); }
    return 0;
}

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    uint32_t *d_out; unsigned long long *d_clk;
    CHK(hipMalloc((void **)&d_out, (size_t)n_cu * 8 * 256 * 4));
    CHK(hipMalloc((void **)&d_clk, (size_t)n_cu * 8 * 8 * 8));
    std::string json = "{\"device\": \"" + std::string(prop.gcnArchName) + "\", \"cus\": " + std::to_string(n_cu) + ", \"mixes\": [\n";
    if (run_all<0>(n_cu, d_out, d_clk, json)) return 1;
    json += "\n]}\n";
    fputs(json.c_str(), stdout);
    return 0;
}
