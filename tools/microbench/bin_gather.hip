// bin_gather.hip — would position-binned gathers beat k_align's direct random gathers?  
//
// k_align's scan issues one 16-byte gather per candidate at a random position of the 1.55 GB packed reference: 1.2 G candidates per 2^20 pairs at
// the chip's random-request rate (52 G/s: 23 ms).  The same gathers from an L2-resident window run at 240 G/s (profiles/r02e_probe_sweep.json).
// The proposal: scatter (position, tag) records into 1 024 position bins (1.5 MB of reference each), then gather bin by bin with the bin's slice
// resident in one XCD's L2, eight XCDs on eight bins.  What the naive version hides: the candidate is compared with a READ, and the read's words
// have to be there, too — they travel in the record (24-byte records), or the reads are processed in sub-batches small enough for their packed
// words to stay in L2 beside the slice (2^15 pairs = 5 MB do not fit a 4 MB L2 with it; 2^14 pairs = 2.6 MB) with the bins re-walked per sub-batch.
//
// Timed here, N records with uniformly random positions (N = 2^30):
//   direct          one pass: record i gathers 16 bytes at its position (+ 16 bytes of "read words" at a random read of a 168 MB table)
//   binned8         scatter 8-byte records {position, tag} into bins, then per bin: gather 16 bytes from the slice; the read words gathered
//                   from the 168 MB table by tag (random again)
//   binned24        the records carry 16 bytes of read words: scatter 24-byte records, per bin gather the reference only
//   subbatch        64 sub-batches of N / 64 records: each scattered and gathered on its own with its 2.6 MB read table (L2-resident gathers
//                   by tag) — the reference slices are fetched once per sub-batch
// Every variant writes one word per record that passes a 1/16 filter (survivors are few) to out[tag] and keeps a checksum alive.
//
// build: hipcc -O3 --offload-arch=gfx950 -o bin_gather bin_gather.hip ; run on the GPU box: ./bin_gather > bin_gather.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef uint32_t u4 __attribute__((ext_vector_type(4), aligned(4)));
static constexpr uint32_t N_BINS = 1024;
static constexpr uint64_t REF_BYTES = 1552ull << 20;                    // the packed reference (both strand copies)
static constexpr uint32_t BIN_BYTES = (uint32_t)(REF_BYTES / N_BINS);   // 1.5 MB
static constexpr uint32_t READ_TABLE = 168u << 20;                      // 2^21 reads x 80 bytes
static constexpr uint32_t SUB_TABLE = (1u << 15) * 80u;                 // 2^14 pairs' reads: 2.6 MB

__device__ __forceinline__ uint32_t rnd(uint32_t i, uint32_t salt) { uint32_t x = i * 2654435761u + salt; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16; return x; }
__device__ __forceinline__ uint32_t rd_off(uint32_t i, uint32_t table_bytes) { return (uint32_t)(((uint64_t)rnd(i, 7u) * (table_bytes - 16)) >> 32) & ~15u; }   // the read's 16 bytes in a table of that size
__device__ __forceinline__ uint32_t pos_of(uint32_t i) { return (uint32_t)(((uint64_t)rnd(i, 1u) * (REF_BYTES - 64)) >> 32) & ~3u; }   // 4-byte aligned byte offset, as the packed copy's gathers

__global__ __launch_bounds__(256) void k_fill(uint32_t *p, size_t n_words, uint32_t salt) { for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n_words; i += (size_t)gridDim.x * 256) p[i] = rnd((uint32_t)i, salt); }
__global__ __launch_bounds__(256) void k_positions(uint32_t *pos, uint32_t n) { for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) pos[i] = pos_of(i); }

// direct: what k_align does today
__global__ __launch_bounds__(256) void k_direct(const uint8_t *ref, const uint8_t *reads, uint32_t read_mask, const uint32_t *pos, uint32_t n, uint32_t *out)
{
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const u4 r = *reinterpret_cast<const u4 *>(ref + pos[i]);
        const u4 q = *reinterpret_cast<const u4 *>(reads + rd_off(i, read_mask));
        const uint32_t v = (r.x ^ q.x) + (r.y ^ q.y) + (r.z ^ q.z) + (r.w ^ q.w);
        if ((v & 15u) == 0) out[i] = v;
    }
}

// scatter pass.  A block takes tiles of TILE records: bins counted in LDS (a record's rank inside its bin of the tile is the counter's old value),
// one global atomic per non-empty bin and tile reserves the tile's run in the bin, the records go out into those runs (12 records = 96 / 288 bytes per
// bin and tile on average: neighbouring stores of a run merge in L2).  REC: 2 words {position, tag} or 6 {position, tag, 4 read words}.
template <int REC>
__global__ __launch_bounds__(1024) void k_scatter(const uint32_t *pos, const uint8_t *reads, uint32_t read_mask, uint32_t first, uint32_t n, uint32_t *cursor /* [N_BINS] */,
                                                   const uint32_t *bin_start /* [N_BINS] */, uint32_t *rec)
{
    constexpr uint32_t PER = 12, TILE = 1024 * PER;
    __shared__ uint32_t hist[N_BINS], base[N_BINS];
    for (uint32_t t0 = blockIdx.x * TILE; t0 < n; t0 += gridDim.x * TILE) {
        hist[threadIdx.x] = 0;
        __syncthreads();
        uint32_t p[PER], rk[PER];
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t i = t0 + k * 1024 + threadIdx.x;
            p[k] = i < n ? pos[first + i] : 0xffffffffu;
            rk[k] = i < n ? atomicAdd(&hist[p[k] / BIN_BYTES], 1u) : 0u;
        }
        __syncthreads();
        { const uint32_t h = hist[threadIdx.x]; base[threadIdx.x] = h ? bin_start[threadIdx.x] + atomicAdd(&cursor[threadIdx.x], h) : 0u; }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t i = t0 + k * 1024 + threadIdx.x;
            if (i >= n) continue;
            uint32_t *dst = rec + (size_t)(base[p[k] / BIN_BYTES] + rk[k]) * REC;
            dst[0] = p[k]; dst[1] = first + i;
            if (REC == 6) { const u4 q = *reinterpret_cast<const u4 *>(reads + rd_off(first + i, read_mask)); dst[2] = q.x; dst[3] = q.y; dst[4] = q.z; dst[5] = q.w; }
        }
        __syncthreads();
    }
}

// gather pass: the blocks of XCD x (blockIdx % 8: a grid's blocks go to the XCDs round-robin) walk the bins x, x + 8, ... together, so that a bin's
// 1.5 MB slice is fetched into that XCD's L2 once and serves all the bin's records
template <int REC>
__global__ __launch_bounds__(256) void k_bin_gather(const uint8_t *ref, const uint8_t *reads, uint32_t read_mask, const uint32_t *rec, const uint32_t *bin_start, const uint32_t *count, uint32_t *out)
{
    const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3, J = gridDim.x >> 3;
    for (uint32_t b = xcd; b < N_BINS; b += 8) {
        const uint32_t s = bin_start[b], c = count[b];
        for (uint32_t i = j * 256 + threadIdx.x; i < c; i += J * 256) {
            const uint32_t *r_ = rec + (size_t)(s + i) * REC;
            const uint32_t p = r_[0], tag = r_[1];
            const u4 r = *reinterpret_cast<const u4 *>(ref + p);
            u4 q;
            if (REC == 6) { q.x = r_[2]; q.y = r_[3]; q.z = r_[4]; q.w = r_[5]; }
            else q = *reinterpret_cast<const u4 *>(reads + rd_off(tag, read_mask));
            const uint32_t v = (r.x ^ q.x) + (r.y ^ q.y) + (r.z ^ q.z) + (r.w ^ q.w);
            if ((v & 15u) == 0) out[tag] = v;
        }
    }
}

// bins of the records [first, first + n): counts and starts (a counting pass over the positions; part of what a real pipeline pays, timed with the scatter)
__global__ __launch_bounds__(1024) void k_count(const uint32_t *pos, uint32_t first, uint32_t n, uint32_t *count)
{
    __shared__ uint32_t hist[N_BINS];
    hist[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * 1024 + threadIdx.x; i < n; i += gridDim.x * 1024) atomicAdd(&hist[pos[first + i] / BIN_BYTES], 1u);
    __syncthreads();
    if (hist[threadIdx.x]) atomicAdd(&count[threadIdx.x], hist[threadIdx.x]);
}
__global__ __launch_bounds__(1024) void k_starts(const uint32_t *count, uint32_t *bin_start, uint32_t *cursor)
{
    __shared__ uint32_t s[N_BINS];
    s[threadIdx.x] = count[threadIdx.x];
    __syncthreads();
    for (uint32_t o = 1; o < N_BINS; o <<= 1) { const uint32_t v = threadIdx.x >= o ? s[threadIdx.x - o] : 0; __syncthreads(); s[threadIdx.x] += v; __syncthreads(); }
    bin_start[threadIdx.x] = s[threadIdx.x] - count[threadIdx.x];
    cursor[threadIdx.x] = 0;
}

struct Timer {
    hipEvent_t a, b;
    Timer() { (void)hipEventCreate(&a); (void)hipEventCreate(&b); }
    void start() { (void)hipEventRecord(a, 0); }
    float stop() { (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b); float ms = 0; (void)hipEventElapsedTime(&ms, a, b); return ms; }
};

int main(int argc, char **argv)
{
    const uint32_t N = argc > 1 ? (uint32_t)atoll(argv[1]) : (1u << 30);
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    uint8_t *ref, *reads;
    uint32_t *pos, *rec, *out, *count, *bin_start, *cursor;
    CHK(hipMalloc((void **)&ref, REF_BYTES)); CHK(hipMalloc((void **)&reads, READ_TABLE));
    CHK(hipMalloc((void **)&pos, (size_t)N * 4)); CHK(hipMalloc((void **)&rec, (size_t)N * 24)); CHK(hipMalloc((void **)&out, (size_t)N * 4));
    CHK(hipMalloc((void **)&count, N_BINS * 4)); CHK(hipMalloc((void **)&bin_start, N_BINS * 4)); CHK(hipMalloc((void **)&cursor, N_BINS * 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t *)ref, (size_t)(REF_BYTES / 4), 11u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t *)reads, (size_t)(READ_TABLE / 4), 13u);
    hipLaunchKernelGGL(k_positions, dim3(8192), dim3(256), 0, 0, pos, N);
    CHK(hipMemset(out, 0, (size_t)N * 4));
    CHK(hipDeviceSynchronize());
    Timer T;
    const int grid = n_cu * 8;
    auto binned = [&](int recw, uint32_t first, uint32_t n, uint32_t read_mask, float &t_scatter, float &t_gather) {
        T.start();
        (void)hipMemsetAsync(count, 0, N_BINS * 4, 0);
        hipLaunchKernelGGL(k_count, dim3(n_cu * 2), dim3(1024), 0, 0, pos, first, n, count);
        hipLaunchKernelGGL(k_starts, dim3(1), dim3(1024), 0, 0, count, bin_start, cursor);
        if (recw == 2) hipLaunchKernelGGL(k_scatter<2>, dim3(n_cu), dim3(1024), 0, 0, pos, reads, read_mask, first, n, cursor, bin_start, rec);
        else hipLaunchKernelGGL(k_scatter<6>, dim3(n_cu), dim3(1024), 0, 0, pos, reads, read_mask, first, n, cursor, bin_start, rec);
        t_scatter += T.stop();
        T.start();
        if (recw == 2) hipLaunchKernelGGL(k_bin_gather<2>, dim3(grid), dim3(256), 0, 0, ref, reads, read_mask, rec, bin_start, count, out);
        else hipLaunchKernelGGL(k_bin_gather<6>, dim3(grid), dim3(256), 0, 0, ref, reads, read_mask, rec, bin_start, count, out);
        t_gather += T.stop();
    };
    // warm-up of each kernel on a small range
    { float a = 0, b = 0; hipLaunchKernelGGL(k_direct, dim3(grid), dim3(256), 0, 0, ref, reads, READ_TABLE, pos, 1u << 22, out); binned(2, 0, 1u << 22, READ_TABLE, a, b); binned(6, 0, 1u << 22, READ_TABLE, a, b); }
    CHK(hipDeviceSynchronize());
    float t_direct[2] = {1e30f, 1e30f}, t8[2] = {0, 0}, t24[2] = {0, 0}, tsub[2] = {0, 0}, t_direct_ref_only = 1e30f;
    for (int rep = 0; rep < 2; rep++) {
        T.start();
        hipLaunchKernelGGL(k_direct, dim3(grid), dim3(256), 0, 0, ref, reads, READ_TABLE, pos, N, out);
        t_direct[0] = std::min(t_direct[0], T.stop());
        T.start();   // (the reads' words from an L2-resident table: the reference gather alone is random)
        hipLaunchKernelGGL(k_direct, dim3(grid), dim3(256), 0, 0, ref, reads, 1u << 20, pos, N, out);
        t_direct_ref_only = std::min(t_direct_ref_only, T.stop());
    }
    binned(2, 0, N, READ_TABLE, t8[0], t8[1]);
    binned(6, 0, N, READ_TABLE, t24[0], t24[1]);
    const uint32_t SUBS = 64;
    for (uint32_t s = 0; s < SUBS; s++) binned(2, (uint32_t)((uint64_t)N * s / SUBS), (uint32_t)((uint64_t)N * (s + 1) / SUBS - (uint64_t)N * s / SUBS), SUB_TABLE, tsub[0], tsub[1]);
    CHK(hipDeviceSynchronize());
    uint32_t chk = 0;
    CHK(hipMemcpy(&chk, out + 12345, 4, hipMemcpyDeviceToHost));
    const double n = (double)N;
    printf("{\"device\": \"%s\", \"cus\": %d, \"records\": %u, \"bins\": %u, \"bin_bytes\": %u, \"reference_bytes\": %llu,\n", prop.gcnArchName, n_cu, N, N_BINS, BIN_BYTES, (unsigned long long)REF_BYTES);
    printf(" \"direct\": {\"ms\": %.2f, \"G_records_per_s\": %.1f, \"note\": \"two random 16-byte gathers per record (reference, and the read's words in a 168 MB table)\"},\n", t_direct[0], n / t_direct[0] / 1e6);
    printf(" \"direct_reads_in_l2\": {\"ms\": %.2f, \"G_records_per_s\": %.1f, \"note\": \"the reference gather alone is random (read words from a 1 MB table): k_align's case, its read lives in registers\"},\n", t_direct_ref_only, n / t_direct_ref_only / 1e6);
    printf(" \"binned_8_byte_records\": {\"scatter_ms\": %.2f, \"gather_ms\": %.2f, \"total_ms\": %.2f, \"note\": \"the reference gather hits the bin's slice in L2, the read words are a random gather again\"},\n", t8[0], t8[1], t8[0] + t8[1]);
    printf(" \"binned_24_byte_records\": {\"scatter_ms\": %.2f, \"gather_ms\": %.2f, \"total_ms\": %.2f, \"note\": \"16 bytes of read words travel in the record\"},\n", t24[0], t24[1], t24[0] + t24[1]);
    printf(" \"sub_batches\": {\"n\": %u, \"scatter_ms\": %.2f, \"gather_ms\": %.2f, \"total_ms\": %.2f, \"note\": \"8-byte records, read words from a 2.6 MB table per sub-batch, every bin slice fetched once per sub-batch\"},\n", SUBS, tsub[0], tsub[1], tsub[0] + tsub[1]);
    printf(" \"ratio_to_direct_reads_in_l2\": {\"binned_8\": %.2f, \"binned_24\": %.2f, \"sub_batches\": %.2f}, \"checksum\": %u}\n", (t8[0] + t8[1]) / t_direct_ref_only, (t24[0] + t24[1]) / t_direct_ref_only, (tsub[0] + tsub[1]) / t_direct_ref_only, chk);
    return 0;
}
