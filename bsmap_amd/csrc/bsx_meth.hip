// bsx_meth.hip — methylation-ratio pile-up on the GPU (reference: methratio.py of the BSMAP tree).
//
// The reference walks the alignments one by one in Python: duplicate removal by fragment end (methratio.py:50-54),
// fill-in trimming (:55-63), then for every reference C (G on the minus strand) under the read one or two counter
// increments (:104-114), optionally folding CpG pairs (:118-128), and a table of the covered cytosines (:135-151).
// Here the per-alignment part runs as one wave per alignment with atomic counters in HBM:
//   k_meth_first   input-order duplicate removal: atomicMin of the alignment's global index on its fragment-end slot
//   k_meth_pile    trim, bounds test, compare read and reference letters, atomicAdd on depth / methylated counters
//   k_meth_cpg     fold the G of every CG into its C
//   k_meth_count / k_meth_emit   ordered compaction of the positions the table will list
// The host side (bsmap_amd/methratio.py) parses the option surface, the FASTA and the BSP / SAM lines, and prints the
// table with the reference's arithmetic.  All counters are integers: results are exact.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "bsx_cpus.h"
#include "bsx_internal.h"

namespace {

typedef unsigned long long u64;

struct MethDev {
    const uint8_t *ref;      // all chromosomes, upper-case letters, concatenated
    const u64 *chr_off;      // [n_chr+1] offsets into ref / depth / meth
    uint32_t *depth, *meth;  // per reference position
    uint32_t *first;         // [2][total] fragment-end slots of the duplicate filter (direction 1, 2), or null
    u64 total;
    uint32_t n_chr;
};

struct AlnBatch {
    const uint32_t *chr;
    const int64_t *pos;        // 0-based leftmost reference position of the (untrimmed) read
    const uint8_t *strand;     // 0 '++', 1 '-+', 2 '+-', 3 '--'   (first char: reference strand, second: read orientation)
    const int32_t *insert;     // BSP column 8 / SAM TLEN
    const int64_t *cut_at;     // SAM with insert > 0: PNEXT-1 (the read is cut where its mate starts); -1 otherwise
    const uint8_t *seq;
    const u64 *seq_off;
    uint32_t n, index_base, trim_fillin;
};

// Python's s[a:b] on a string of length n (a, b may be negative or past the ends) as [lo, hi)
__device__ __forceinline__ void py_slice(int64_t n, bool has_a, int64_t a, bool has_b, int64_t b, int64_t &lo, int64_t &hi)
{
    lo = 0; hi = n;
    if (has_a) { if (a < 0) a += n; lo = a < 0 ? 0 : (a > n ? n : a); }
    if (has_b) { if (b < 0) b += n; hi = b < 0 ? 0 : (b > n ? n : b); }
    if (hi < lo) hi = lo;
}

__device__ __forceinline__ bool dup_slot(const MethDev &M, const AlnBatch &B, uint32_t i, u64 &slot)
{
    const uint32_t c = B.chr[i];
    const int64_t len = (int64_t)(B.seq_off[i + 1] - B.seq_off[i]), clen = (int64_t)(M.chr_off[c + 1] - M.chr_off[c]);
    const uint32_t st = B.strand[i];
    const bool end_side = st == 2 || st == 1;  // '+-' or '-+': the fragment end is the read's right end (methratio.py:51)
    const int64_t fe = end_side ? B.pos[i] + len : B.pos[i];
    if (fe < 0 || fe >= clen) return false;    // (the reference would index outside its coverage array here)
    slot = (end_side ? M.total : 0) + M.chr_off[c] + (u64)fe;
    return true;
}

__global__ void k_meth_first(MethDev M, AlnBatch B)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B.n) return;
    u64 slot;
    if (dup_slot(M, B, i, slot)) atomicMin(&M.first[slot], B.index_base + i);
}

// one wave per alignment
__global__ __launch_bounds__(256) void k_meth_pile(MethDev M, AlnBatch B, u64 *n_valid)
{
    const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= B.n) return;
    if (M.first) {
        u64 slot;
        if (dup_slot(M, B, i, slot) && M.first[slot] != B.index_base + i) return;  // an earlier alignment owns this fragment end
    }
    const uint32_t c = B.chr[i], st = B.strand[i];
    const int64_t n0 = (int64_t)(B.seq_off[i + 1] - B.seq_off[i]), clen = (int64_t)(M.chr_off[c + 1] - M.chr_off[c]);
    int64_t pos = B.pos[i], lo = 0, hi = n0;  // the read letters that survive are seq[lo:hi), aligned at pos
    const int64_t t = (int64_t)B.trim_fillin, ins = B.insert[i];
    if (t > 0) {  // methratio.py:55-63
        if (st == 2) py_slice(n0, false, 0, true, -t, lo, hi);                         // '+-': seq[:-t]
        else if (st == 3) { py_slice(n0, true, t, false, 0, lo, hi); pos += t; }       // '--': seq[t:], pos + t
        else if (ins != 0 && n0 > (ins < 0 ? -ins : ins) - t) {
            const int64_t trim_nt = n0 - ((ins < 0 ? -ins : ins) - t);
            if (st == 0) py_slice(n0, false, 0, true, -trim_nt, lo, hi);               // '++': seq[:-trim_nt]
            else if (st == 1) { py_slice(n0, true, trim_nt, false, 0, lo, hi); pos += trim_nt; }  // '-+': seq[trim_nt:]
        }
    }
    if (B.cut_at[i] >= 0) {  // SAM, insert > 0: seq[:PNEXT-1-pos]  (methratio.py:64)
        int64_t l2, h2;
        py_slice(hi - lo, false, 0, true, B.cut_at[i] - pos, l2, h2);
        hi = lo + h2;
    }
    const int64_t len = hi - lo;
    if (pos + len > clen) return;  // methratio.py:100
    if (lane == 0) atomicAdd(n_valid, 1ull);
    if (pos < 0) return;           // (a negative position would make the reference slice from the chromosome's end: never produced by bsmap)
    const uint8_t match = (st & 1) ? 'G' : 'C', convert = (st & 1) ? 'A' : 'T';  // strand[0]: '+' -> C/T, '-' -> G/A
    const uint8_t *s = B.seq + B.seq_off[i] + lo;
    const u64 g0 = M.chr_off[c] + (u64)pos;
    for (int64_t k = lane; k < len; k += 64) {
        if (M.ref[g0 + k] != match) continue;
        const uint8_t ch = s[k];
        if (ch == convert) atomicAdd(&M.depth[g0 + k], 1u);
        else if (ch == match) { atomicAdd(&M.meth[g0 + k], 1u); atomicAdd(&M.depth[g0 + k], 1u); }
    }
}

__global__ void k_meth_cpg(MethDev M)
{
    // every "CG" of every chromosome (occurrences cannot overlap): the G's counters move onto the C  (methratio.py:118-128)
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g + 1 < M.total; g += (u64)gridDim.x * blockDim.x) {
        if (M.ref[g] != 'C' || M.ref[g + 1] != 'G') continue;
        // (a C that is the last letter of a chromosome must not pair with the first letter of the next one)
        uint32_t lo = 0, hi = M.n_chr;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (M.chr_off[mid] <= g) lo = mid; else hi = mid; }
        if (g + 1 >= M.chr_off[lo + 1]) continue;
        M.depth[g] += M.depth[g + 1]; M.meth[g] += M.meth[g + 1];
        M.depth[g + 1] = 0; M.meth[g + 1] = 0;
    }
}

// table rows of one chromosome, in position order: blocks of 1024 positions are counted, scanned, and emitted
__global__ __launch_bounds__(256) void k_meth_count(MethDev M, u64 g0, u64 n, uint32_t min_depth, int meth0, uint32_t *blk_rows, u64 *nc_nd)
{
    __shared__ uint32_t s_rows;
    __shared__ u64 s_nc, s_nd;
    if (threadIdx.x == 0) { s_rows = 0; s_nc = 0; s_nd = 0; }
    __syncthreads();
    uint32_t rows = 0; u64 nc = 0, nd = 0;
    for (int r = 0; r < 4; r++) {
        const u64 k = (u64)blockIdx.x * 1024 + (u64)r * 256 + threadIdx.x;
        if (k >= n) continue;
        const uint32_t d = M.depth[g0 + k], m = M.meth[g0 + k];
        if (d < min_depth) continue;
        nc++; nd += d;
        if (m != 0 || meth0) rows++;
    }
    atomicAdd(&s_rows, rows); atomicAdd(&s_nc, nc); atomicAdd(&s_nd, nd);
    __syncthreads();
    if (threadIdx.x == 0) { blk_rows[blockIdx.x] = s_rows; if (s_nc) { atomicAdd(&nc_nd[0], s_nc); atomicAdd(&nc_nd[1], s_nd); } }
}

__global__ __launch_bounds__(256) void k_meth_emit(MethDev M, u64 g0, u64 n, uint32_t min_depth, int meth0, const uint32_t *blk_start, uint32_t *out_pos,
                                                   uint32_t *out_depth, uint32_t *out_meth, u64 *out_ctx)
{
    __shared__ uint32_t s_wave[4];
    uint32_t base = blk_start[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int r = 0; r < 4; r++) {  // positions blockIdx*1024 + r*256 + thread: ascending with (r, thread)
        const u64 k = (u64)blockIdx.x * 1024 + (u64)r * 256 + threadIdx.x;
        uint32_t d = 0, m = 0;
        bool row = false;
        if (k < n) { d = M.depth[g0 + k]; m = M.meth[g0 + k]; row = d >= min_depth && (m != 0 || meth0); }
        const u64 bal = __ballot(row);
        if (lane == 0) s_wave[wv] = (uint32_t)__builtin_popcountll(bal);
        __syncthreads();
        uint32_t off = base;
        for (int w = 0; w < wv; w++) off += s_wave[w];
        if (row) {
            const uint32_t o = off + (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1));
            out_pos[o] = (uint32_t)k; out_depth[o] = d; out_meth[o] = m;
            if (out_ctx) {  // refcr[i-2:i+3] with Python's slice rules (empty for i < 2 on any chromosome longer than five letters), byte 7 = the letter at i
                u64 ctx = 0;
                int nb = 0;
                if (k >= 2) for (u64 q = k - 2; q < k + 3 && q < n; q++) ctx |= (u64)M.ref[g0 + q] << (8 * nb++);
                out_ctx[o] = ctx | ((u64)M.ref[g0 + k] << 56);
            }
        }
        base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
}

}  // namespace

struct bsx_meth {
    int device = 0;
    uint32_t n_chr = 0;
    std::vector<u64> chr_off;
    uint8_t *d_ref = nullptr;
    u64 *d_chr_off = nullptr, *d_counts = nullptr;  // d_counts: [0] valid alignments, [1] nc, [2] nd
    uint32_t *d_depth = nullptr, *d_meth = nullptr, *d_first = nullptr;
    uint32_t index_base = 0;
    hipStream_t stream = nullptr;
    // report buffers of the last bsx_meth_report_chr
    uint32_t *d_blk = nullptr, *d_blk_start = nullptr, *d_out[3] = {nullptr, nullptr, nullptr};
    u64 *d_ctx = nullptr;
    size_t blk_cap = 0, out_cap = 0;
    void *d_temp = nullptr; size_t temp_cap = 0;
    uint32_t last_rows = 0;
    std::vector<std::string> names;  // filled by bsx_meth_create_from_fasta
    MethDev dev() const { MethDev M; M.ref = d_ref; M.chr_off = d_chr_off; M.depth = d_depth; M.meth = d_meth; M.first = d_first; M.total = chr_off.back(); M.n_chr = n_chr; return M; }
};

extern "C" void bsx_meth_destroy(bsx_meth *m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    for (void *q : {(void *)m->d_ref, (void *)m->d_chr_off, (void *)m->d_counts, (void *)m->d_depth, (void *)m->d_meth, (void *)m->d_first, (void *)m->d_blk,
                    (void *)m->d_blk_start, (void *)m->d_out[0], (void *)m->d_out[1], (void *)m->d_out[2], (void *)m->d_ctx, m->d_temp})
        if (q) (void)hipFree(q);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

extern "C" int bsx_meth_create(uint32_t n_chr, const uint64_t *chr_len, int rm_dup, int device, bsx_meth **out)
{
    if (!n_chr || !chr_len || !out) return BSX_ERR_ARG;
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) { g_bsx_err = "no HIP device visible; libbsx has no CPU fallback"; return BSX_ERR_NODEVICE; }
    if (device < 0 || device >= nd) return BSX_ERR_NODEVICE;
    HIP_TRY(hipSetDevice(device));
    bsx_meth *m = new bsx_meth();
    m->device = device; m->n_chr = n_chr;
    m->chr_off.assign(1, 0);
    for (uint32_t c = 0; c < n_chr; c++) m->chr_off.push_back(m->chr_off.back() + chr_len[c]);
    const u64 total = m->chr_off.back();
    auto fail = [&](int rc) { bsx_meth_destroy(m); return rc; };
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) return fail(BSX_ERR_DEVICE);
    if (hipMalloc((void **)&m->d_ref, total + 16) != hipSuccess || hipMalloc((void **)&m->d_depth, (total + 16) * 4) != hipSuccess ||
        hipMalloc((void **)&m->d_meth, (total + 16) * 4) != hipSuccess || hipMalloc((void **)&m->d_chr_off, (n_chr + 1) * 8) != hipSuccess ||
        hipMalloc((void **)&m->d_counts, 64) != hipSuccess)
        return fail(BSX_ERR_NOMEM);
    if (rm_dup && hipMalloc((void **)&m->d_first, 2 * (total + 16) * 4) != hipSuccess) return fail(BSX_ERR_NOMEM);
    if (hipMemsetAsync(m->d_ref, 0, total + 16, m->stream) != hipSuccess || hipMemsetAsync(m->d_depth, 0, (total + 16) * 4, m->stream) != hipSuccess ||
        hipMemsetAsync(m->d_meth, 0, (total + 16) * 4, m->stream) != hipSuccess || hipMemsetAsync(m->d_counts, 0, 64, m->stream) != hipSuccess)
        return fail(BSX_ERR_DEVICE);
    if (rm_dup && hipMemsetAsync(m->d_first, 0xff, 2 * (total + 16) * 4, m->stream) != hipSuccess) return fail(BSX_ERR_DEVICE);
    if (hipMemcpyAsync(m->d_chr_off, m->chr_off.data(), (n_chr + 1) * 8, hipMemcpyHostToDevice, m->stream) != hipSuccess) return fail(BSX_ERR_DEVICE);
    if (hipStreamSynchronize(m->stream) != hipSuccess) return fail(BSX_ERR_DEVICE);
    *out = m;
    return BSX_OK;
}

extern "C" int bsx_meth_set_reference(bsx_meth *m, uint32_t chr, const char *upper_seq)
{
    if (!m || chr >= m->n_chr || !upper_seq) return BSX_ERR_ARG;
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipMemcpy(m->d_ref + m->chr_off[chr], upper_seq, m->chr_off[chr + 1] - m->chr_off[chr], hipMemcpyHostToDevice));
    return BSX_OK;
}

extern "C" int bsx_meth_add(bsx_meth *m, uint32_t n, const uint32_t *chr, const int64_t *pos, const uint8_t *strand, const int32_t *insert, const int64_t *cut_at,
                            const char *seqs, const uint64_t *seq_off, uint32_t trim_fillin)
{
    if (!m || (n && (!chr || !pos || !strand || !insert || !cut_at || !seqs || !seq_off))) return BSX_ERR_ARG;
    if (!n) return BSX_OK;
    if ((u64)m->index_base + n >= 0xFFFFFFFFull) return BSX_ERR_LIMIT;
    for (uint32_t i = 0; i < n; i++) if (chr[i] >= m->n_chr || strand[i] > 3) return BSX_ERR_ARG;
    HIP_TRY(hipSetDevice(m->device));
    const u64 nbytes = seq_off[n];
    uint32_t *d_chr = nullptr; int64_t *d_pos = nullptr, *d_cut = nullptr; uint8_t *d_strand = nullptr, *d_seq = nullptr; int32_t *d_ins = nullptr; u64 *d_off = nullptr;
    int rc = BSX_OK;
    auto chk = [&](hipError_t e) { if (e != hipSuccess && rc == BSX_OK) rc = bsx_hip_fail(e, "bsx_meth_add", __FILE__, __LINE__); };
    chk(hipMalloc((void **)&d_chr, (size_t)n * 4)); chk(hipMalloc((void **)&d_pos, (size_t)n * 8)); chk(hipMalloc((void **)&d_cut, (size_t)n * 8));
    chk(hipMalloc((void **)&d_strand, n)); chk(hipMalloc((void **)&d_ins, (size_t)n * 4)); chk(hipMalloc((void **)&d_seq, nbytes + 16)); chk(hipMalloc((void **)&d_off, ((size_t)n + 1) * 8));
    if (rc == BSX_OK) {
        chk(hipMemcpyAsync(d_chr, chr, (size_t)n * 4, hipMemcpyHostToDevice, m->stream)); chk(hipMemcpyAsync(d_pos, pos, (size_t)n * 8, hipMemcpyHostToDevice, m->stream));
        chk(hipMemcpyAsync(d_cut, cut_at, (size_t)n * 8, hipMemcpyHostToDevice, m->stream)); chk(hipMemcpyAsync(d_strand, strand, n, hipMemcpyHostToDevice, m->stream));
        chk(hipMemcpyAsync(d_ins, insert, (size_t)n * 4, hipMemcpyHostToDevice, m->stream)); chk(hipMemcpyAsync(d_seq, seqs, nbytes, hipMemcpyHostToDevice, m->stream));
        chk(hipMemcpyAsync(d_off, seq_off, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, m->stream));
    }
    if (rc == BSX_OK) {
        AlnBatch B; B.chr = d_chr; B.pos = d_pos; B.strand = d_strand; B.insert = d_ins; B.cut_at = d_cut; B.seq = d_seq; B.seq_off = d_off;
        B.n = n; B.index_base = m->index_base; B.trim_fillin = trim_fillin;
        const MethDev M = m->dev();
        if (m->d_first) hipLaunchKernelGGL(k_meth_first, dim3((n + 255) / 256), dim3(256), 0, m->stream, M, B);
        hipLaunchKernelGGL(k_meth_pile, dim3((n + 3) / 4), dim3(256), 0, m->stream, M, B, m->d_counts);
        chk(hipGetLastError());
        chk(hipStreamSynchronize(m->stream));
        m->index_base += n;
    }
    for (void *q : {(void *)d_chr, (void *)d_pos, (void *)d_cut, (void *)d_strand, (void *)d_ins, (void *)d_seq, (void *)d_off}) if (q) (void)hipFree(q);
    return rc;
}

extern "C" int bsx_meth_combine_cpg(bsx_meth *m)
{
    if (!m) return BSX_ERR_ARG;
    HIP_TRY(hipSetDevice(m->device));
    hipLaunchKernelGGL(k_meth_cpg, dim3(4096), dim3(256), 0, m->stream, m->dev());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(m->stream));
    return BSX_OK;
}

extern "C" int bsx_meth_valid_mappings(bsx_meth *m, uint64_t *n)
{
    if (!m || !n) return BSX_ERR_ARG;
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipMemcpy(n, m->d_counts, 8, hipMemcpyDeviceToHost));
    return BSX_OK;
}

extern "C" int bsx_meth_report_chr(bsx_meth *m, uint32_t chr, uint32_t min_depth, int meth0, uint32_t *n_rows, uint64_t *n_covered, uint64_t *sum_depth)
{
    if (!m || chr >= m->n_chr || !n_rows) return BSX_ERR_ARG;
    HIP_TRY(hipSetDevice(m->device));
    const u64 g0 = m->chr_off[chr], n = m->chr_off[chr + 1] - g0;
    const size_t nblk = (size_t)((n + 1023) / 1024);
    *n_rows = 0; m->last_rows = 0;
    if (n_covered) *n_covered = 0;
    if (sum_depth) *sum_depth = 0;
    if (!nblk) return BSX_OK;
    if (nblk + 1 > m->blk_cap) {
        if (m->d_blk) (void)hipFree(m->d_blk);
        if (m->d_blk_start) (void)hipFree(m->d_blk_start);
        m->d_blk = m->d_blk_start = nullptr; m->blk_cap = 0;
        HIP_TRY(hipMalloc((void **)&m->d_blk, (nblk + 1) * 4)); HIP_TRY(hipMalloc((void **)&m->d_blk_start, (nblk + 1) * 4));
        m->blk_cap = nblk + 1;
    }
    HIP_TRY(hipMemsetAsync(m->d_counts + 1, 0, 16, m->stream));
    HIP_TRY(hipMemsetAsync(m->d_blk + nblk, 0, 4, m->stream));
    const MethDev M = m->dev();
    hipLaunchKernelGGL(k_meth_count, dim3((unsigned)nblk), dim3(256), 0, m->stream, M, g0, n, min_depth, meth0, m->d_blk, m->d_counts + 1);
    HIP_TRY(hipGetLastError());
    size_t need = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, need, m->d_blk, m->d_blk_start, 0u, nblk + 1, rocprim::plus<uint32_t>(), m->stream));
    if (need > m->temp_cap) { if (m->d_temp) (void)hipFree(m->d_temp); m->d_temp = nullptr; m->temp_cap = 0; HIP_TRY(hipMalloc(&m->d_temp, need)); m->temp_cap = need; }
    HIP_TRY(rocprim::exclusive_scan(m->d_temp, need, m->d_blk, m->d_blk_start, 0u, nblk + 1, rocprim::plus<uint32_t>(), m->stream));
    uint32_t rows = 0; u64 cnt[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(&rows, m->d_blk_start + nblk, 4, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(cnt, m->d_counts + 1, 16, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (rows > m->out_cap) {
        for (int k = 0; k < 3; k++) { if (m->d_out[k]) (void)hipFree(m->d_out[k]); m->d_out[k] = nullptr; }
        if (m->d_ctx) (void)hipFree(m->d_ctx);
        m->d_ctx = nullptr;
        m->out_cap = 0;
        for (int k = 0; k < 3; k++) HIP_TRY(hipMalloc((void **)&m->d_out[k], (size_t)rows * 4));
        HIP_TRY(hipMalloc((void **)&m->d_ctx, (size_t)rows * 8));
        m->out_cap = rows;
    }
    if (rows) {
        hipLaunchKernelGGL(k_meth_emit, dim3((unsigned)nblk), dim3(256), 0, m->stream, M, g0, n, min_depth, meth0, m->d_blk_start, m->d_out[0], m->d_out[1], m->d_out[2], m->d_ctx);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(m->stream));
    }
    *n_rows = rows; m->last_rows = rows;
    if (n_covered) *n_covered = cnt[0];
    if (sum_depth) *sum_depth = cnt[1];
    return BSX_OK;
}

extern "C" int bsx_meth_fetch_rows(bsx_meth *m, uint32_t *pos, uint32_t *depth, uint32_t *meth)
{
    if (!m || !pos || !depth || !meth) return BSX_ERR_ARG;
    HIP_TRY(hipSetDevice(m->device));
    if (!m->last_rows) return BSX_OK;
    HIP_TRY(hipMemcpy(pos, m->d_out[0], (size_t)m->last_rows * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(depth, m->d_out[1], (size_t)m->last_rows * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(meth, m->d_out[2], (size_t)m->last_rows * 4, hipMemcpyDeviceToHost));
    return BSX_OK;
}

// ---- host side for whole files: the reference's per-line Python is the slow part of the tool once the pile-up is on the GPU ----
namespace {

struct ParsedChunk {
    std::vector<uint32_t> chr; std::vector<int64_t> pos, cut; std::vector<uint8_t> strand; std::vector<int32_t> insert;
    std::vector<char> seq; std::vector<u64> off{0};
    u64 lines = 0;
    int bad = 0;
};

inline bool field(const char *&p, const char *e, const char *&b, size_t &n)  // next tab-separated column of the line [p, e)
{
    if (p > e) return false;
    b = p;
    const char *t = (const char *)memchr(p, '\t', (size_t)(e - p));
    if (!t) { n = (size_t)(e - p); p = e + 1; }
    else { n = (size_t)(t - p); p = t + 1; }
    return true;
}

// get_alignment's filters (methratio.py:31-48) for the lines of [b, e)
void parse_chunk(const char *b, const char *e, int sam, const std::unordered_map<std::string, uint32_t> &cid, int unique, int pair, ParsedChunk &o)
{
    std::string key;
    while (b < e) {
        const char *nl = (const char *)memchr(b, '\n', (size_t)(e - b));
        const char *le = nl ? nl : e;  // line without its newline
        const char *p = b;
        b = nl ? nl + 1 : e;
        o.lines++;
        const char *c[12]; size_t n[12];
        int nc = 0;
        while (nc < 12 && field(p, le, c[nc], n[nc])) nc++;
        if (sam) {
            if (n[0] && c[0][0] == '@') { o.lines--; continue; }
            if (nc < 11) { o.bad = 1; continue; }
            const long flag = strtol(std::string(c[1], n[1]).c_str(), nullptr, 10);
            if ((flag & 0x4) || (unique && (flag & 0x100)) || (pair && !(flag & 0x2))) continue;
            key.assign(c[2], n[2]);
            auto it = cid.find(key);
            if (it == cid.end()) continue;
            const long long pos = strtoll(std::string(c[3], n[3]).c_str(), nullptr, 10) - 1, insert = strtoll(std::string(c[8], n[8]).c_str(), nullptr, 10);
            // strand from the first ZS:Z: tag among the optional fields
            const char *q = c[10] + n[10] + 1;
            int st = -1;
            while (q <= le) {
                const char *fb; size_t fn;
                if (!field(q, le, fb, fn)) break;
                if (fn >= 7 && memcmp(fb, "ZS:Z:", 5) == 0) { st = (fb[5] == '-' ? 1 : 0) | (fb[6] == '-' ? 2 : 0); break; }
            }
            if (st < 0) { o.bad = 2; continue; }
            o.chr.push_back(it->second); o.pos.push_back(pos); o.strand.push_back((uint8_t)st); o.insert.push_back((int32_t)insert);
            o.cut.push_back(insert > 0 ? strtoll(std::string(c[7], n[7]).c_str(), nullptr, 10) - 1 : -1);
            o.seq.insert(o.seq.end(), c[9], c[9] + n[9]); o.off.push_back(o.seq.size());
        } else {
            if (nc < 4) { o.bad = 1; continue; }
            const char f0 = n[3] > 0 ? c[3][0] : 0, f1 = n[3] > 1 ? c[3][1] : 0;
            if ((f0 == 'N' && f1 == 'M') || (f0 == 'Q' && f1 == 'C')) continue;
            if (unique && !(f0 == 'U' && f1 == 'M')) continue;
            if (nc < 8) { o.bad = 1; continue; }
            if (pair && n[7] == 1 && c[7][0] == '0') continue;
            key.assign(c[4], n[4]);
            auto it = cid.find(key);
            if (it == cid.end()) continue;
            if (n[6] < 2) { o.bad = 2; continue; }
            o.chr.push_back(it->second); o.pos.push_back(strtoll(std::string(c[5], n[5]).c_str(), nullptr, 10) - 1);
            o.strand.push_back((uint8_t)((c[6][0] == '-' ? 1 : 0) | (c[6][1] == '-' ? 2 : 0)));
            o.insert.push_back((int32_t)strtoll(std::string(c[7], n[7]).c_str(), nullptr, 10)); o.cut.push_back(-1);
            o.seq.insert(o.seq.end(), c[1], c[1] + n[1]); o.off.push_back(o.seq.size());
        }
    }
}

// BAM, streamed: the BGZF blocks are inflated (in parallel) a bounded window at a time — a whole-genome BAM is hundreds of GB
// inflated — and the alignment records of each window are handed to `flush` in file order (first-wins duplicate removal
// depends on it); a record, or the header, that straddles the window edge is carried into the next window.  Fields as
// `samtools view -X` would print them (what the reference reads): flag bits, RNAME from the header, POS, PNEXT, TLEN, SEQ, ZS:Z
template <class Flush>
int stream_bam(const char *base, size_t len, const std::unordered_map<std::string, uint32_t> &cid, int unique, int pair, size_t window, u64 &lines, int &bad_records,
               Flush &&flush)
{
    struct Blk { size_t in_off, in_len, out_len; };
    std::vector<Blk> blks;
    size_t p = 0;
    const unsigned char *u = (const unsigned char *)base;
    while (p + 18 <= len) {
        if (u[p] != 0x1f || u[p + 1] != 0x8b || u[p + 2] != 8 || !(u[p + 3] & 4)) return 1;
        const unsigned xlen = u[p + 10] | (u[p + 11] << 8);
        unsigned bsize = 0;
        for (unsigned x = 0; x + 4 <= xlen;) {
            const unsigned char *f = u + p + 12 + x;
            const unsigned slen = f[2] | (f[3] << 8);
            if (f[0] == 'B' && f[1] == 'C' && slen == 2) bsize = (f[4] | (f[5] << 8)) + 1;
            x += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || p + bsize > len) return 1;
        const unsigned isize = u[p + bsize - 4] | (u[p + bsize - 3] << 8) | (u[p + bsize - 2] << 16) | ((unsigned)u[p + bsize - 1] << 24);
        blks.push_back(Blk{p + 12 + xlen, bsize - 12 - xlen - 8, isize});
        p += bsize;
    }
    auto rd32 = [](const unsigned char *q) { int32_t v; memcpy(&v, q, 4); return v; };
    static const char nt16[] = "=ACMGRSVTWYHKDBN";
    std::vector<unsigned char> buf;
    std::vector<int64_t> ref_id;  // BAM reference index -> our chromosome id or -1
    int32_t n_ref = 0;
    bool header_done = false;
    size_t carry = 0;
    for (size_t bi = 0; bi < blks.size();) {
        size_t bj = bi, out = 0;
        while (bj < blks.size() && (bj == bi || out + blks[bj].out_len <= window)) { out += blks[bj].out_len; bj++; }
        buf.resize(carry + out + 8);
        {
            std::vector<size_t> at(bj - bi + 1, carry);
            for (size_t i = bi; i < bj; i++) at[i - bi + 1] = at[i - bi] + blks[i].out_len;
            std::atomic<size_t> next(bi);
            std::atomic<int> bad(0);
            auto work = [&] {
                for (size_t i; (i = next.fetch_add(1)) < bj;) {
                    if (!blks[i].out_len) continue;
                    z_stream z;
                    memset(&z, 0, sizeof(z));
                    if (inflateInit2(&z, -15) != Z_OK) { bad = 1; continue; }
                    z.next_in = const_cast<unsigned char *>(u + blks[i].in_off); z.avail_in = (unsigned)blks[i].in_len;
                    z.next_out = buf.data() + at[i - bi]; z.avail_out = (unsigned)blks[i].out_len;
                    if (inflate(&z, Z_FINISH) != Z_STREAM_END) bad = 1;
                    inflateEnd(&z);
                }
            };
            const size_t nt = std::min<size_t>(std::max<size_t>(1, (bj - bi) / 16), std::max(1u, std::min(32u, bsx_usable_cpus())));
            std::vector<std::thread> th;
            for (size_t t = 1; t < nt; t++) th.emplace_back(work);
            work();
            for (std::thread &x : th) x.join();
            if (bad) return 1;
        }
        const unsigned char *b = buf.data(), *e = b + carry + out, *q = b;
        bi = bj;
        if (!header_done) {  // magic, text, reference names; all of it must be inside the buffer before it is read
            bool complete = e - b >= 12 && e - b >= 12 + (ptrdiff_t)rd32(b + 4);
            if (e - b >= 4 && memcmp(b, "BAM\1", 4) != 0) return 1;
            if (complete) {
                q = b + 8 + rd32(b + 4);
                n_ref = rd32(q); q += 4;
                ref_id.clear();
                for (int32_t r = 0; r < n_ref && complete; r++) {
                    if (q + 4 > e) { complete = false; break; }
                    const int32_t ln = rd32(q);
                    if (ln < 1) return 1;
                    if (q + 4 + ln + 4 > e) { complete = false; break; }
                    auto it = cid.find(std::string((const char *)q + 4, strnlen((const char *)q + 4, (size_t)ln)));
                    ref_id.push_back(it == cid.end() ? -1 : (int64_t)it->second);
                    q += 4 + ln + 4;
                }
            }
            if (!complete) { carry += out; continue; }  // header longer than the window so far: read on
            header_done = true;
        }
        ParsedChunk o;
        while (q + 4 <= e) {
            const int32_t bs = rd32(q);
            if (bs < 32) return 1;
            if (q + 4 + bs > e) break;  // the rest of this record is in the next window
            const unsigned char *r = q + 4;
            q += 4 + (size_t)bs;
            o.lines++;
            const int32_t tid = rd32(r), pos = rd32(r + 4), l_seq = rd32(r + 16), npos = rd32(r + 24), tlen = rd32(r + 28);
            const unsigned l_name = r[8], n_cigar = r[12] | (r[13] << 8), flag = r[14] | (r[15] << 8);
            if ((flag & 0x4) || (unique && (flag & 0x100)) || (pair && !(flag & 0x2))) continue;
            if (tid < 0 || tid >= n_ref || ref_id[(size_t)tid] < 0) continue;
            const unsigned char *sq = r + 32 + l_name + 4 * n_cigar, *ql = sq + ((size_t)l_seq + 1) / 2, *aux = ql + l_seq;
            if (l_seq < 0 || aux > r + bs) return 1;
            int st = -1;
            while (aux + 3 <= r + bs) {  // optional fields: find ZS:Z
                const unsigned char t0 = aux[0], t1 = aux[1], ty = aux[2];
                aux += 3;
                if (ty == 'Z' || ty == 'H') {
                    const size_t n = strnlen((const char *)aux, (size_t)(r + bs - aux));
                    if (t0 == 'Z' && t1 == 'S' && ty == 'Z' && n >= 2) { st = (aux[0] == '-' ? 1 : 0) | (aux[1] == '-' ? 2 : 0); break; }
                    aux += n + 1;
                } else if (ty == 'A' || ty == 'c' || ty == 'C') aux += 1;
                else if (ty == 's' || ty == 'S') aux += 2;
                else if (ty == 'i' || ty == 'I' || ty == 'f') aux += 4;
                else if (ty == 'B') { if (aux + 5 > r + bs) break; const unsigned char sub = aux[0]; const int32_t cnt = rd32(aux + 1); aux += 5 + (size_t)cnt * ((sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4); }
                else break;
            }
            if (st < 0) { bad_records = 2; continue; }
            o.chr.push_back((uint32_t)ref_id[(size_t)tid]); o.pos.push_back(pos); o.strand.push_back((uint8_t)st); o.insert.push_back(tlen);
            o.cut.push_back(tlen > 0 ? (int64_t)npos : -1);
            for (int32_t i = 0; i < l_seq; i++) o.seq.push_back(nt16[(sq[i >> 1] >> ((~i & 1) << 2)) & 0xf]);
            o.off.push_back(o.seq.size());
        }
        lines += o.lines;
        if (!o.chr.empty()) { const int rc = flush(o); if (rc) return rc; }
        carry = (size_t)(e - q);
        if (carry) memmove(buf.data(), q, carry);
    }
    if (!header_done || carry) return 1;  // no header, or a record cut off by the end of the file
    return 0;
}

}  // namespace

extern "C" int bsx_meth_add_file(bsx_meth *m, const char *path, int sam, const char *const *chr_names, int unique, int pair, uint32_t trim_fillin, uint64_t *n_lines)
{
    if (!m || !path || (!chr_names && m->names.size() != m->n_chr)) return BSX_ERR_ARG;
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) { g_bsx_err = std::string("cannot open ") + path; return BSX_ERR_IO; }
    struct stat st;
    fstat(fd, &st);
    const size_t len = (size_t)st.st_size;
    if (n_lines) *n_lines = 0;
    if (!len) { ::close(fd); return BSX_OK; }
    const char *base = (const char *)mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (base == MAP_FAILED) { g_bsx_err = std::string("cannot map ") + path; return BSX_ERR_IO; }
    std::unordered_map<std::string, uint32_t> cid;
    for (uint32_t c = 0; c < m->n_chr; c++) cid.emplace(chr_names ? std::string(chr_names[c]) : m->names[c], c);
    if (sam == 2) {  // BAM, a window of inflated blocks at a time (BSX_BAM_WINDOW bytes, default 256 MB)
        u64 lines = 0;
        int bad = 0, rc_add = BSX_OK;
        const size_t window = getenv("BSX_BAM_WINDOW") ? (size_t)std::max(1ll, atoll(getenv("BSX_BAM_WINDOW"))) : ((size_t)256 << 20);
        const int rcs = stream_bam(base, len, cid, unique, pair, window, lines, bad, [&](ParsedChunk &pc) {
            if (pc.chr.size() > 0xffffffffull) { rc_add = BSX_ERR_LIMIT; return 1; }
            pc.seq.push_back(0);
            rc_add = bsx_meth_add(m, (uint32_t)pc.chr.size(), pc.chr.data(), pc.pos.data(), pc.strand.data(), pc.insert.data(), pc.cut.data(), pc.seq.data(),
                                  (const uint64_t *)pc.off.data(), trim_fillin);
            return rc_add != BSX_OK ? 1 : 0;
        });
        munmap((void *)base, len);
        if (n_lines) *n_lines = lines;
        if (rc_add != BSX_OK) return rc_add;
        if (rcs) { g_bsx_err = std::string("not a readable BAM file: ") + path; return BSX_ERR_IO; }
        if (bad == 2) { g_bsx_err = "alignment record without strand information"; return BSX_ERR_ARG; }
        return BSX_OK;
    }
    // pieces of ~256 MB (bounded host memory), each cut into per-thread chunks at line starts; alignments keep the file's order
    const size_t piece = 256u << 20;
    int rc = BSX_OK;
    for (size_t p0 = 0; p0 < len && rc == BSX_OK;) {
        size_t p1 = std::min(len, p0 + piece);
        if (p1 < len) { const char *nl = (const char *)memchr(base + p1, '\n', len - p1); p1 = nl ? (size_t)(nl - base) + 1 : len; }
        const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(32u, std::max(1u, bsx_usable_cpus())), (p1 - p0) / (1u << 20) + 1));
        std::vector<size_t> cut(nt + 1, p1);
        cut[0] = p0;
        for (unsigned t = 1; t < nt; t++) {
            const size_t g = p0 + (p1 - p0) * t / nt;
            const char *nl = (const char *)memchr(base + g, '\n', p1 - g);
            cut[t] = nl ? (size_t)(nl - base) + 1 : p1;
        }
        std::vector<ParsedChunk> pc(nt);
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) th.emplace_back([&, t] { parse_chunk(base + cut[t], base + cut[t + 1], sam, cid, unique, pair, pc[t]); });
        parse_chunk(base + cut[0], base + cut[1], sam, cid, unique, pair, pc[0]);
        for (std::thread &x : th) x.join();
        ParsedChunk all;
        for (ParsedChunk &q : pc) {
            if (q.bad == 2) { g_bsx_err = "alignment line without strand information"; rc = BSX_ERR_ARG; }
            if (n_lines) *n_lines += q.lines;
            const u64 b0 = all.seq.size();
            all.chr.insert(all.chr.end(), q.chr.begin(), q.chr.end()); all.pos.insert(all.pos.end(), q.pos.begin(), q.pos.end());
            all.cut.insert(all.cut.end(), q.cut.begin(), q.cut.end()); all.strand.insert(all.strand.end(), q.strand.begin(), q.strand.end());
            all.insert.insert(all.insert.end(), q.insert.begin(), q.insert.end()); all.seq.insert(all.seq.end(), q.seq.begin(), q.seq.end());
            for (size_t i = 1; i < q.off.size(); i++) all.off.push_back(b0 + q.off[i]);
        }
        if (rc == BSX_OK && !all.chr.empty()) {
            all.seq.push_back(0);
            rc = bsx_meth_add(m, (uint32_t)all.chr.size(), all.chr.data(), all.pos.data(), all.strand.data(), all.insert.data(), all.cut.data(), all.seq.data(), (const uint64_t *)all.off.data(), trim_fillin);
        }
        p0 = p1;
    }
    munmap((void *)base, len);
    return rc;
}

// the table of methratio.py:130-151 for the chromosomes in `order` (the reference sorts the names), same arithmetic and formats
extern "C" int bsx_meth_write_table(bsx_meth *m, const char *path, uint32_t n_order, const uint32_t *order, const char *const *chr_names, uint32_t min_depth, int meth0,
                                    uint64_t *n_covered, uint64_t *sum_depth)
{
    if (!m || !path || (!chr_names && m->names.size() != m->n_chr)) return BSX_ERR_ARG;
    std::vector<uint32_t> sorted_order;
    if (!order) {  // the reference writes the chromosomes in sorted name order
        for (uint32_t c = 0; c < m->n_chr; c++) sorted_order.push_back(c);
        std::sort(sorted_order.begin(), sorted_order.end(), [&](uint32_t a, uint32_t b) {
            return (chr_names ? std::string(chr_names[a]) : m->names[a]) < (chr_names ? std::string(chr_names[b]) : m->names[b]); });
        order = sorted_order.data(); n_order = m->n_chr;
    }
    FILE *f = fopen(path, "w");
    if (!f) { g_bsx_err = std::string("cannot write ") + path; return BSX_ERR_IO; }
    fputs("chr\tpos\tstrand\tcontext\tratio\ttotal_C\tmethy_C\tCI_lower\tCI_upper\n", f);
    u64 nc = 0, nd = 0;
    int rc = BSX_OK;
    const double z95 = 1.96, z95sq = 1.96 * 1.96;
    std::vector<uint32_t> pos, dep, met; std::vector<u64> ctx;
    std::string buf;
    for (uint32_t k = 0; k < n_order && rc == BSX_OK; k++) {
        const uint32_t c = order[k];
        if (c >= m->n_chr) { rc = BSX_ERR_ARG; break; }
        uint32_t rows = 0; uint64_t cov = 0, sd_ = 0;
        rc = bsx_meth_report_chr(m, c, min_depth, meth0, &rows, &cov, &sd_);
        if (rc) break;
        nc += cov; nd += sd_;
        if (!rows) continue;
        pos.resize(rows); dep.resize(rows); met.resize(rows); ctx.resize(rows);
        rc = bsx_meth_fetch_rows(m, pos.data(), dep.data(), met.data());
        if (rc) break;
        if (hipMemcpy(ctx.data(), m->d_ctx, (size_t)rows * 8, hipMemcpyDeviceToHost) != hipSuccess) { rc = BSX_ERR_DEVICE; break; }
        const char *name = chr_names ? chr_names[c] : m->names[c].c_str();
        // rows are formatted by a pool of threads, each a contiguous range, and written in order
        const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(32u, std::max(1u, bsx_usable_cpus())), rows / 65536 + 1));
        std::vector<std::string> parts(nt);
        auto fmt = [&](unsigned t) {
            std::string &o = parts[t];
            const size_t lo = (size_t)rows * t / nt, hi = (size_t)rows * (t + 1) / nt;
            o.reserve((hi - lo) * 56);
            char line[256];
            for (size_t i = lo; i < hi; i++) {
                const double d = dep[i], mm = met[i];
                const double ratio = mm / d;
                const double pmid = ratio + z95sq / (2 * d);
                const double sd = z95 * pow(ratio * (1 - ratio) / d + z95sq / (4 * d * d), 0.5);
                const double norminator = 1 + z95sq / d;
                char cx[8]; int nb = 0;
                for (; nb < 5; nb++) { const char ch = (char)((ctx[i] >> (8 * nb)) & 0xff); if (!ch) break; cx[nb] = ch; }
                cx[nb] = 0;
                const char letter = (char)(ctx[i] >> 56);
                const int w = snprintf(line, sizeof(line), "%s\t%u\t%c\t%s\t%.3f\t%u\t%u\t%.3f\t%.3f\n", name, pos[i] + 1, letter == 'C' ? '+' : '-', cx, ratio, dep[i], met[i],
                                       (pmid - sd) / norminator, (pmid + sd) / norminator);
                o.append(line, (size_t)w);
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) th.emplace_back(fmt, t);
        fmt(0);
        for (std::thread &x : th) x.join();
        for (const std::string &o : parts) fwrite(o.data(), 1, o.size(), f);
    }
    fclose(f);
    if (n_covered) *n_covered = nc;
    if (sum_depth) *sum_depth = nd;
    return rc;
}

// methratio.py:67-77 on a memory map: a record starts at a line whose first character is '>', its name is the first
// token of line[1:-1], its sequence the concatenation of the stripped lines, upper-cased; `chroms_csv` is the -c filter
extern "C" int bsx_meth_create_from_fasta(const char *path, const char *chroms_csv, int rm_dup, int device, bsx_meth **out)
{
    if (!path || !out) return BSX_ERR_ARG;
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) { g_bsx_err = std::string("cannot open ") + path; return BSX_ERR_IO; }
    struct stat st;
    fstat(fd, &st);
    const size_t len = (size_t)st.st_size;
    if (!len) { ::close(fd); return BSX_ERR_IO; }
    const char *base = (const char *)mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (base == MAP_FAILED) return BSX_ERR_IO;
    std::vector<std::string> want;
    if (chroms_csv && *chroms_csv) { std::string t(chroms_csv); size_t a = 0; for (;;) { const size_t c = t.find(',', a); want.push_back(t.substr(a, c == std::string::npos ? c : c - a)); if (c == std::string::npos) break; a = c + 1; } }
    auto wanted = [&](const std::string &n) { if (want.empty()) return true; for (const std::string &w : want) if (w == n) return true; return false; };
    struct Rec { std::string name; size_t b, e; };
    std::vector<Rec> recs;
    {   // header lines
        size_t p = 0;
        std::string cur; bool have = false; size_t sb = 0;
        while (p < len) {
            const char *nl = (const char *)memchr(base + p, '\n', len - p);
            const size_t le = nl ? (size_t)(nl - base) + 1 : len;  // line including its newline
            if (base[p] == '>') {
                if (have && wanted(cur)) recs.push_back(Rec{cur, sb, p});
                // name = line[1:-1].split()[0]
                size_t a = p + 1, z = le > p + 1 ? le - 1 : p + 1;
                while (a < z && (base[a] == ' ' || (base[a] >= '\t' && base[a] <= '\r'))) a++;
                size_t q = a;
                while (q < z && !(base[q] == ' ' || (base[q] >= '\t' && base[q] <= '\r'))) q++;
                cur.assign(base + a, q - a); have = true; sb = le;
                p = le;
                continue;
            }
            // jump to the next header line
            const char *g = p < len ? (const char *)memmem(base + p, len - p, "\n>", 2) : nullptr;
            p = g ? (size_t)(g - base) + 1 : len;
        }
        if (have && wanted(cur)) recs.push_back(Rec{cur, sb, len});
    }
    // (a name given twice keeps its last record, as the reference's dict does)
    for (size_t i = 0; i < recs.size(); i++) for (size_t j = i + 1; j < recs.size(); j++) if (recs[i].name == recs[j].name) { recs.erase(recs.begin() + (long)i); i--; break; }
    if (recs.empty()) { munmap((void *)base, len); g_bsx_err = "no sequence selected from the reference file"; return BSX_ERR_ARG; }
    std::vector<std::vector<char>> seqs(recs.size());
    {
        std::atomic<size_t> next(0);
        auto work = [&] {
            for (size_t i; (i = next.fetch_add(1)) < recs.size();) {
                std::vector<char> &o = seqs[i];
                o.reserve(recs[i].e - recs[i].b);
                size_t p = recs[i].b;
                while (p < recs[i].e) {
                    const char *nl = (const char *)memchr(base + p, '\n', recs[i].e - p);
                    size_t a = p, z = nl ? (size_t)(nl - base) : recs[i].e;
                    p = nl ? (size_t)(nl - base) + 1 : recs[i].e;
                    while (a < z && (base[a] == ' ' || (base[a] >= '\t' && base[a] <= '\r'))) a++;     // line.strip()
                    while (z > a && (base[z - 1] == ' ' || (base[z - 1] >= '\t' && base[z - 1] <= '\r'))) z--;
                    const size_t o0 = o.size();
                    o.insert(o.end(), base + a, base + z);
                    for (size_t k = o0; k < o.size(); k++) if (o[k] >= 'a' && o[k] <= 'z') o[k] = (char)(o[k] - 32);
                }
            }
        };
        const size_t nt = std::min<size_t>(recs.size(), std::max(1u, std::min(32u, bsx_usable_cpus())));
        std::vector<std::thread> th;
        for (size_t t = 1; t < nt; t++) th.emplace_back(work);
        work();
        for (std::thread &x : th) x.join();
    }
    munmap((void *)base, len);
    std::vector<uint64_t> lens;
    for (const std::vector<char> &q : seqs) lens.push_back(q.size());
    bsx_meth *m = nullptr;
    int rc = bsx_meth_create((uint32_t)recs.size(), lens.data(), rm_dup, device, &m);
    if (rc) return rc;
    for (size_t i = 0; i < recs.size() && rc == BSX_OK; i++) {
        m->names.push_back(recs[i].name);
        if (!seqs[i].empty() && hipMemcpy(m->d_ref + m->chr_off[i], seqs[i].data(), seqs[i].size(), hipMemcpyHostToDevice) != hipSuccess) rc = BSX_ERR_DEVICE;
        std::vector<char>().swap(seqs[i]);
    }
    if (rc) { bsx_meth_destroy(m); return rc; }
    *out = m;
    return BSX_OK;
}

extern "C" uint32_t bsx_meth_n_chr(const bsx_meth *m) { return m ? m->n_chr : 0; }
extern "C" const char *bsx_meth_chr_name(const bsx_meth *m, uint32_t c) { return (m && c < m->names.size()) ? m->names[c].c_str() : ""; }
