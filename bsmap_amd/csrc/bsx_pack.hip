// bsx_pack.hip — RefSeq::Run_ConvertBinseq on the device (round 6): the FASTA text is uploaded and packed there.
//
// Reference behaviour restated (file:line in BSMAP v2.6): RefSeq::LoadNextSeq / BinSeq / cBinSeq / UnmaskRegion / Run_ConvertBinseq, dbseq.cpp:18-142,215-282.
//
// The host packer (bsx_host.cpp: bsx_pack_fasta) streams 3.1 GB of text through the box's CPUs — 0.9-1.1 s on 16 of them at hg38 size — then assembles the two
// strand copies (1.5 GB, serial) and uploads them from pageable memory: 2.3 s of a run whose mapping phase is 1.8 s.  Here the host only finds the records (the
// '>' bytes, collected by a pool of threads) and the text goes up as it is; the rest is three kernels:
//   k_fa_check : is a record LINE-REGULAR — every line of L characters followed by one '\n', the last line shorter, no other blank anywhere?  Then
//                nt i of the record lies at byte  sb + (i / L) (L + 1) + i % L  and nothing has to be compacted.  One irregular record (CR LF ends, blank
//                lines, tabs, lines of unequal length) sends the whole file to the host packer, which applies the reference's token rules to any text.
//   k_fa_pack  : one thread per 16-nt word of both strand copies (dbseq.cpp:58-111): alphabet / rev_alphabet through two 256-byte tables, 'N' behind the end.
//   k_fa_runs  : where N / X stretches begin and end (a handful per chromosome), appended to a list; the host sorts it and runs the reference's two
//                alternating scans of UnmaskRegion over it exactly as the chunk-parallel host packer does (dbseq.cpp:114-142).
// WGBS references only: the RRBS site tables and the {tag, loc} index are assembled on the host from the text (bsx_host.cpp, bsx_index.hip).
// tests: every packed word, anchor and block of the golden sets and of the oracle's own pack (tests/test_gpu_parity.py::test_reference_and_index,
// tests/test_gpu_pack.py: the texts of tests/test_pack_cpu.py in regular and irregular form).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "bsx_cpus.h"
#include "bsx_internal.h"

namespace {

typedef unsigned long long u64;

struct FaRec { u64 sb; u64 rlen; uint32_t L, nt, n_words, w0; };   // text region [sb, sb + rlen) (a trailing '\n' left out), line length, nt, words per strand copy, first word in the concatenation

__device__ __forceinline__ bool fa_ws(unsigned char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

// violations of line regularity in record `rec` (see above): a blank where a letter must be, a letter or another blank where the '\n' must be
__global__ __launch_bounds__(256) void k_fa_check(const unsigned char *__restrict__ text, const FaRec *__restrict__ recs, uint32_t n_rec, uint32_t *__restrict__ bad)
{
    for (uint32_t c = blockIdx.y; c < n_rec; c += gridDim.y) {
        const FaRec R = recs[c];
        uint32_t v = 0;
        for (u64 o = (u64)blockIdx.x * 256 + threadIdx.x; o < R.rlen; o += (u64)gridDim.x * 256) {
            const unsigned char ch = text[R.sb + o];
            const bool at_nl = o % ((u64)R.L + 1) == R.L;
            v += at_nl ? ch != '\n' : fa_ws(ch);
        }
        if (v) atomicAdd(&bad[c], v);
    }
}

__device__ __forceinline__ unsigned char fa_nt(const unsigned char *__restrict__ text, const FaRec &R, uint32_t i)
{
    return i < R.nt ? text[R.sb + (u64)(i / R.L) * (R.L + 1) + i % R.L] : (unsigned char)'N';
}

__global__ __launch_bounds__(256) void k_fa_pack(const unsigned char *__restrict__ text, const FaRec *__restrict__ recs, uint32_t n_rec, const unsigned char *__restrict__ code_f,
                                                 const unsigned char *__restrict__ code_r, uint32_t *__restrict__ fw, uint32_t *__restrict__ rc)
{
    __shared__ unsigned char tf[256], tr[256];
    tf[threadIdx.x] = code_f[threadIdx.x]; tr[threadIdx.x] = code_r[threadIdx.x];
    __syncthreads();
    for (uint32_t c = blockIdx.y; c < n_rec; c += gridDim.y) {
        const FaRec R = recs[c];
        const uint32_t padded = R.n_words * 16u;
        for (uint32_t w = blockIdx.x * 256u + threadIdx.x; w < R.n_words; w += gridDim.x * 256u) {
            uint32_t x = 0, y = 0;
#pragma unroll 4
            for (uint32_t j = 0; j < 16; j++) {
                const uint32_t p = w * 16u + j, q = padded - 1u - p;   // q: the forward position the rc copy shows at index p (dbseq.cpp:86-111)
                x = (x << 2) | tf[fa_nt(text, R, p)];
                y = (y << 2) | tr[fa_nt(text, R, q)];
            }
            fw[R.w0 + w] = x; rc[R.w0 + w] = y;
        }
    }
}

// events: (record << 33 | nt index << 1 | 1 = a stretch of N / X begins here, 0 = it ends here), appended in no particular order
__global__ __launch_bounds__(256) void k_fa_runs(const unsigned char *__restrict__ text, const FaRec *__restrict__ recs, uint32_t n_rec, u64 *__restrict__ events,
                                                 uint32_t cap, uint32_t *__restrict__ n_events)
{
    for (uint32_t c = blockIdx.y; c < n_rec; c += gridDim.y) {
        const FaRec R = recs[c];
        for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i <= R.nt; i += gridDim.x * 256u) {
            auto nx = [&](uint32_t k) { const unsigned char ch = fa_nt(text, R, k); return k < R.nt && (ch == 'N' || ch == 'X' || ch == 'n' || ch == 'x'); };
            const bool cur = nx(i), prev = i ? nx(i - 1) : false;
            if (cur != prev) {
                const uint32_t at = atomicAdd(n_events, 1u);
                if (at < cap) events[at] = ((u64)c << 33) | ((u64)i << 1) | (cur ? 1u : 0u);
            }
        }
    }
}

struct FaHostRec { std::string name; u64 sb, se; };

int nt_index_h(int c)
{
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; }
    return -1;
}
inline bool is_space_h(char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

}  // namespace

// BSX_OK: the reference is packed on the device (r's host tables and d_refcat / d_crefcat are filled, bsx_planes_build is still to be called);
// 1: not applicable (RRBS, an irregular record, too many N / X stretches) — nothing is left behind and the caller takes the host packer;
// < 0: an error.
int bsx_pack_fasta_device(const bsx_params &P, const char *text, uint64_t n, bsx_ref &r)
{
    if (P.rrbs || n < (1u << 16) || getenv("BSX_HOST_PACK")) return 1;   // (small texts: the host packer is instant; BSX_HOST_PACK=1: test hook)
    const bool timing = getenv("BSX_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_0 = now();
    // ---- records: the reference reads with operator>> (dbseq.cpp:18-54) — the first non-blank character is consumed unchecked, the next token is the name,
    // the rest of that line is dropped, then tokens are concatenated until one begins with '>'.  The '>' bytes are collected by a pool of threads.
    std::vector<u64> gt;
    {
        const unsigned nt = std::max(1u, std::min(32u, bsx_usable_cpus()));
        const u64 CH = 8u << 20;
        const size_t nck = (size_t)((n + CH - 1) / CH);
        std::vector<std::vector<u64>> part(nck);
        std::atomic<size_t> next(0);
        auto work = [&] {
            for (size_t k; (k = next.fetch_add(1)) < nck;) {
                const char *p = text + k * CH, *e = text + std::min<u64>(n, (k + 1) * CH);
                while (p < e) { const char *g = (const char *)memchr(p, '>', (size_t)(e - p)); if (!g) break; part[k].push_back((u64)(g - text)); p = g + 1; }
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
        work();
        for (std::thread &t : th) t.join();
        for (auto &v : part) gt.insert(gt.end(), v.begin(), v.end());
    }
    std::vector<FaHostRec> recs;
    {
        u64 i = 0;
        auto skip_ws = [&] { while (i < n && is_space_h(text[i])) i++; };
        for (;;) {
            skip_ws();
            if (i >= n) break;
            i++;   // fin >> c
            skip_ws();
            const u64 b = i;
            while (i < n && !is_space_h(text[i])) i++;
            FaHostRec R;
            R.name.assign(text + b, text + i);
            const char *nl = (const char *)memchr(text + i, '\n', n - i);
            i = nl ? (u64)(nl - text) + 1 : n;
            skip_ws();
            if (i >= n || text[i] == '>') break;   // empty sequence: LoadNextSeq returned 0, loading stops
            R.sb = i;
            // the record ends in front of the next token that starts with '>'
            u64 q = n;
            for (auto it = std::lower_bound(gt.begin(), gt.end(), i); it != gt.end(); ++it) if (is_space_h(text[*it - 1])) { q = *it; break; }
            R.se = q;
            i = q;
            recs.push_back(std::move(R));
        }
    }
    if (recs.empty()) return 1;   // (the host packer reports the empty reference)
    const double t_recs = now();
    // ---- line structure of every record, from its first line; the device checks the rest
    std::vector<FaRec> fr(recs.size());
    uint64_t words = 0;
    for (size_t c = 0; c < recs.size(); c++) {
        const FaHostRec &R = recs[c];
        u64 rlen = R.se - R.sb;
        if (rlen && text[R.se - 1] == '\n') rlen--;
        const char *nl = (const char *)memchr(text + R.sb, '\n', (size_t)rlen);
        const u64 L = nl ? (u64)(nl - (text + R.sb)) : rlen;
        if (L == 0 || L > 0xFFFFFFFEull) return 1;
        const u64 nt = rlen / (L + 1) * L + rlen % (L + 1);
        if (nt >= 0xFFFFFFFFull - 64) return BSX_ERR_LIMIT;
        const uint32_t nw = (uint32_t)((nt + BSX_SEGLEN - 1) / BSX_SEGLEN) + 2;   // BinSeq: two spare words (dbseq.cpp:60)
        if ((words + nw + 2 * BSX_REF_MARGIN) * BSX_SEGLEN >= 0xFFFFFFFFull) return BSX_ERR_LIMIT;
        fr[c] = FaRec{R.sb, rlen, (uint32_t)L, (uint32_t)nt, nw, (uint32_t)(BSX_REF_MARGIN + words)};
        words += nw;
    }
    const uint32_t n_rec = (uint32_t)fr.size();
    uint8_t code_f[256], code_r[256];
    for (int c = 0; c < 256; c++) {
        const int k = nt_index_h(c);
        code_f[c] = P.bit_nt[k < 0 ? 0 : k];          // alphabet[]: unknown -> code of 'A'
        code_r[c] = P.bit_nt[k < 0 ? 3 : 3 - k];      // rev_alphabet[]: unknown -> code of 'T'
    }
    HIP_TRY(hipSetDevice(r.device));
    unsigned char *d_text = nullptr, *d_codes = nullptr;
    FaRec *d_recs = nullptr;
    uint32_t *d_bad = nullptr, *d_nev = nullptr;
    u64 *d_ev = nullptr;
    const uint32_t EV_CAP = 1u << 22;
    auto cleanup = [&] { for (void *q : {(void *)d_text, (void *)d_codes, (void *)d_recs, (void *)d_bad, (void *)d_nev, (void *)d_ev}) if (q) (void)hipFree(q); };
    auto fail = [&](int code) { cleanup(); if (r.d_refcat) { (void)hipFree(r.d_refcat); r.d_refcat = r.d_crefcat = nullptr; } return code; };
    // (no room for the text beside what the device already holds — several lanes on one GPU, a shared device: not an error, the host packer needs none of it)
#define PK_TRY(x) do { hipError_t e_ = (x); if (e_ == hipErrorOutOfMemory) { (void)hipGetLastError(); return fail(1); } if (e_ != hipSuccess) return fail(bsx_hip_fail(e_, #x, __FILE__, __LINE__)); } while (0)
    PK_TRY(hipMalloc((void **)&d_text, n + 64));
    PK_TRY(hipMalloc((void **)&d_codes, 512));
    PK_TRY(hipMalloc((void **)&d_recs, (size_t)n_rec * sizeof(FaRec)));
    PK_TRY(hipMalloc((void **)&d_bad, (size_t)n_rec * 4));
    PK_TRY(hipMalloc((void **)&d_nev, 4));
    PK_TRY(hipMalloc((void **)&d_ev, (size_t)EV_CAP * 8));
    const double t_alloc = now();
    // (one hipMemcpy from the mapping: 9.5 GB/s, 0.33 s at hg38 size.  Page-locked staging buffers filled by six threads with a stream each were tried and are
    //  SLOWER inside the command line — 0.89 s: their hipHostMalloc calls queue behind the gigabytes the command line is page-locking for its ring at that moment)
    PK_TRY(hipMemcpy(d_text, text, n, hipMemcpyHostToDevice));
    const double t_up = now();
    PK_TRY(hipMemcpy(d_codes, code_f, 256, hipMemcpyHostToDevice));
    PK_TRY(hipMemcpy(d_codes + 256, code_r, 256, hipMemcpyHostToDevice));
    PK_TRY(hipMemcpy(d_recs, fr.data(), (size_t)n_rec * sizeof(FaRec), hipMemcpyHostToDevice));
    PK_TRY(hipMemset(d_bad, 0, (size_t)n_rec * 4));
    PK_TRY(hipMemset(d_nev, 0, 4));
    const dim3 grid(1024, std::min<uint32_t>(n_rec, 64));
    hipLaunchKernelGGL(k_fa_check, grid, dim3(256), 0, 0, d_text, d_recs, n_rec, d_bad);
    PK_TRY(hipGetLastError());
    std::vector<uint32_t> bad(n_rec);
    PK_TRY(hipMemcpy(bad.data(), d_bad, (size_t)n_rec * 4, hipMemcpyDeviceToHost));
    for (uint32_t v : bad) if (v) { cleanup(); return 1; }   // an irregular record: the host packer applies the token rules
    r.n_words = words + 2 * BSX_REF_MARGIN;
    // 64 spare words behind each copy; both strand copies in one allocation, the rc copy right behind the forward one (bsx_api.hip: finish_ref_upload)
    PK_TRY(hipMalloc((void **)&r.d_refcat, 2 * (r.n_words + 64) * 4));
    r.d_crefcat = r.d_refcat + r.n_words + 64;
    PK_TRY(hipMemset(r.d_refcat, 0, 2 * (r.n_words + 64) * 4));
    hipLaunchKernelGGL(k_fa_pack, grid, dim3(256), 0, 0, d_text, d_recs, n_rec, d_codes, d_codes + 256, r.d_refcat, r.d_crefcat);
    PK_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_fa_runs, grid, dim3(256), 0, 0, d_text, d_recs, n_rec, d_ev, EV_CAP, d_nev);
    PK_TRY(hipGetLastError());
    uint32_t nev = 0;
    PK_TRY(hipMemcpy(&nev, d_nev, 4, hipMemcpyDeviceToHost));
    if (nev > EV_CAP) return fail(1);   // a reference with millions of N / X stretches: the host packer
    std::vector<u64> ev(nev);
    if (nev) PK_TRY(hipMemcpy(ev.data(), d_ev, (size_t)nev * 8, hipMemcpyDeviceToHost));
    PK_TRY(hipDeviceSynchronize());
    cleanup();
#undef PK_TRY
    const double t_kern = now();
    std::sort(ev.begin(), ev.end());
    // ---- host tables (bsx_pack_fasta's, from the same quantities)
    r.n_chr = 0; r.sum_length = 0;
    r.anchor.assign(1, BSX_REF_MARGIN * BSX_SEGLEN); r.chr_size.clear(); r.rc_offset.clear(); r.names.clear(); r.blocks.clear(); r.sites.clear(); r.ccgg_index.clear();
    size_t ei = 0;
    for (uint32_t c = 0; c < n_rec; c++) {
        const FaRec &F = fr[c];
        const uint32_t L = F.nt;
        const u64 padded = (u64)F.n_words * BSX_SEGLEN;
        r.names.push_back(recs[c].name); r.chr_size.push_back(L); r.rc_offset.push_back((uint32_t)padded);
        r.anchor.push_back((uint32_t)(((u64)F.w0 + F.n_words) * BSX_SEGLEN));
        r.n_chr++; r.sum_length += L;
        // the N / X stretches of [0, padded): those of the text, and the padding behind its end (joined to a stretch that runs up to the end)
        std::vector<std::pair<u64, u64>> runs;
        for (; ei < ev.size() && (ev[ei] >> 33) == c; ei++) {
            const u64 pos = (ev[ei] >> 1) & 0xFFFFFFFFull;
            if (ev[ei] & 1) runs.emplace_back(pos, (u64)L); else runs.back().second = pos;
        }
        if (!runs.empty() && runs.back().second == L) runs.back().second = padded; else runs.emplace_back((u64)L, padded);
        auto ch_at = [&](u64 i) { return (unsigned char)text[F.sb + i / F.L * ((u64)F.L + 1) + i % F.L]; };
        auto is_acgt = [&](u64 i) { return i < L && nt_index_h(ch_at(i)) >= 0; };
        // UnmaskRegion (dbseq.cpp:114-142): to the next ACGT letter, then to the next N / X — the stretches are skipped whole (bsx_host.cpp: pack_record)
        size_t ri = 0;
        uint32_t begin, end = 0;
        while (end < L) {
            u64 q = end;
            for (;;) {
                while (ri < runs.size() && runs[ri].second <= q) ri++;
                if (ri < runs.size() && runs[ri].first <= q) { q = runs[ri].second; continue; }
                if (q >= padded || is_acgt(q)) break;
                q++;
            }
            if (q >= padded || q > L) break;
            begin = (uint32_t)q;
            while (ri < runs.size() && runs[ri].second <= q) ri++;
            q = ri < runs.size() ? std::max<u64>(q, runs[ri].first) : padded;
            end = q <= L ? (uint32_t)q : L;
            if (end - begin < 30) continue;
            r.blocks.push_back(Block{2 * c, begin, end});
            r.blocks.push_back(Block{2 * c + 1, (uint32_t)padded - end, (uint32_t)padded - begin});
        }
    }
    std::sort(r.blocks.begin(), r.blocks.end(), [](const Block &a, const Block &b) { return a.id < b.id || (a.id == b.id && a.begin < b.begin); });
    if (timing) fprintf(stderr, "{\"device_pack_s\": {\"records\": %.3f, \"alloc\": %.3f, \"upload\": %.3f, \"kernels_and_free\": %.3f, \"blocks\": %.3f}, \"text_bytes\": %llu, \"n_events\": %u}\n",
                        t_recs - t_0, t_alloc - t_recs, t_up - t_alloc, t_kern - t_up, now() - t_kern, (unsigned long long)n, nev);
    return BSX_OK;
}
