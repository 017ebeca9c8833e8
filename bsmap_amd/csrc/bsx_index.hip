// bsx_index.hip — seed index construction on the GPU.
//
// Stands in for RefSeq::CreateIndex (reference dbseq.cpp:516-539): InitialIndex (:308), t_CalKmerFreq_ab (:327),
// AllocIndex (:365) and t_CreateIndex_ab (:409).  The reference makes one heap array per non-empty bucket,
// [2+N, 2+N_fwd, fwd positions ascending..., rc positions ascending...] (dbseq.cpp:381-382,464-465,477-478).
// Here the same content is one CSR: bucket_off[3^S+1], bucket_nfwd[3^S], entries[] — identical entry order.
//
// Method (device): enumerate every sampled position in the reference's visiting order (all forward-copy blocks,
// then all rc-copy blocks, each ascending), tag it with sort key 2*hash+strand, run ONE stable LSD radix sort
// (key,position) and read bucket boundaries back with a binary search per bucket.  A stable sort on the key keeps
// the visiting order inside each (bucket,strand) group, which is exactly the reference's fill order.
// The radix sort is rocPRIM's (library primitive, not on the timed path); everything else is hand-written.
#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>

#include "bsx_internal.h"
#include "bsx_dev.h"

namespace {

struct BlockTab {
    const uint64_t *prefix;   // [n+1] first entry ordinal of each block in visiting order
    const uint32_t *first;    // [n] first sampled position (chr-local)
    const uint32_t *word0;    // [n] word index of the chromosome start inside refcat/crefcat
    const uint32_t *anchor;   // [n] global nt coordinate of the chromosome start
    const uint8_t *strand;    // [n]
    uint32_t n;
};

__global__ void k_enumerate(BlockTab t, uint64_t total, const uint32_t *__restrict__ refcat, const uint32_t *__restrict__ crefcat,
                            uint32_t seed_size, uint32_t interval, uint32_t seed_bits, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t lo = 0, hi = t.n;  // last block with prefix <= e
        while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (t.prefix[mid] <= e) lo = mid; else hi = mid; }
        const uint32_t loc = t.first[lo] + (uint32_t)(e - t.prefix[lo]) * interval;
        const uint32_t *m = (t.strand[lo] ? crefcat : refcat) + t.word0[lo] + (loc >> 4);
        const uint64_t v = ((uint64_t)m[0] << 32) | m[1];
        const uint32_t x = (uint32_t)(v >> (64 - 2 * seed_size - 2 * (loc & 15))) & seed_bits;  // s_MakeSeed_1, dbseq.cpp:286-291
        keys[e] = bsx_seed_hash(x) * 2 + t.strand[lo];
        vals[e] = t.anchor[lo] + loc;                                                            // hit2int, dbseq.cpp:570
    }
}

// plane copy of one strand copy: pair g = {low bits, high bits} of nt [32 g, 32 g + 32) (bsx_dev.h)
__global__ void k_planes(const uint32_t *__restrict__ words, uint64_t n_words, uint32_t *__restrict__ planes, uint64_t n_pairs)
{
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_pairs; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t a = 2 * g < n_words ? words[2 * g] : 0u, b = 2 * g + 1 < n_words ? words[2 * g + 1] : 0u;
        planes[2 * g] = bsx_plane_word(a, b, 0);
        planes[2 * g + 1] = bsx_plane_word(a, b, 1);
    }
}

// bucket_off[k] = first ordinal with key >= 2k ; bucket_nfwd[k] = (first ordinal with key >= 2k+1) - bucket_off[k]
__global__ void k_boundaries(const uint32_t *__restrict__ skeys, uint64_t total, uint32_t n_buckets, uint32_t *__restrict__ bucket_off,
                             uint32_t *__restrict__ bucket_nfwd)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= n_buckets; k += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t lo = 0, hi = total;  // lower_bound(2k)
        const uint32_t want = (uint32_t)(2 * k);
        while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (skeys[mid] < want) lo = mid + 1; else hi = mid; }
        bucket_off[k] = (uint32_t)lo;
        if (k < n_buckets) {
            uint64_t lo2 = lo, hi2 = total;
            while (lo2 < hi2) { uint64_t mid = (lo2 + hi2) >> 1; if (skeys[mid] < want + 1) lo2 = mid + 1; else hi2 = mid; }
            bucket_nfwd[k] = (uint32_t)(lo2 - lo);
        }
    }
}

// The context of every index entry: the 32 reference nt left and right of its seed, as four packed words (16 nt each, first nt in the top bits, like the
// reference copies themselves): {[e - 32, e - 16), [e - 16, e), [e + 16, e + 32), [e + 32, e + 48)} of the strand copy the entry lies on (the low bit of its
// sort key).  The main kernel compares a read's own flanks of the seed with them before it gathers anything (wave_scan_range<.., CTX>): data that arrives
// with the coalesced entry load.  16 bytes per entry: 23.6 GB at hg38 size — what 288 GB of HBM are for.
__global__ void k_context(const uint32_t *__restrict__ skeys, const uint32_t *__restrict__ entries, uint64_t total, const uint32_t *__restrict__ refcat,
                          const uint32_t *__restrict__ crefcat, uint32_t *__restrict__ ctx)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t e = entries[i];
        const uint32_t *m = (skeys[i] & 1u) ? crefcat : refcat;
        uint32_t out[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t x = e + (j < 2 ? 16u * j - 32u : 16u * j - 16u), w = x >> 4, sh = x & 15u;   // (every entry lies behind the 400-word margin: e >= 6400)
            out[j] = sh ? __builtin_amdgcn_alignbit(m[w], m[w + 1], 32u - 2u * sh) : m[w];
        }
        reinterpret_cast<uint4 *>(ctx)[i] = make_uint4(out[0], out[1], out[2], out[3]);
    }
}

template <class T> struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { return hipMalloc(&p, (n ? n : 1) * sizeof(T)) == hipSuccess ? 0 : -1; }
    void reset() { if (p) (void)hipFree(p); p = nullptr; }
    T *release() { T *q = p; p = nullptr; return q; }
};

}  // namespace

// Plane copy of the packed reference for the scan kernels of the heavy pipeline (k_hscan, k_hscan_shared): 2 bits per nt like
// the packed copy (2 x 0.77 GB at hg38 size).  BSX_PLANE_PAD pairs behind each strand copy: a candidate's last gather reaches
// six pairs from its first.
#define BSX_PLANE_PAD 16
int bsx_planes_build(bsx_ref *r)
{
    const uint64_t n_pairs = (r->n_words + 1) / 2 + BSX_PLANE_PAD;
    if (n_pairs * 16 >= 0xFFFFFFFFull) { g_bsx_err = "reference too long for 32-bit plane offsets"; return BSX_ERR_LIMIT; }
    if (r->d_refplane) { (void)hipFree(r->d_refplane); r->d_refplane = nullptr; }
    HIP_TRY(hipMalloc((void **)&r->d_refplane, 2 * n_pairs * 8));
    r->plane_rc_off = (uint32_t)(n_pairs * 8);
    const int grid = (int)std::min<uint64_t>((n_pairs + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(k_planes, dim3(grid), dim3(256), 0, 0, r->d_refcat, r->n_words, r->d_refplane, n_pairs);
    hipLaunchKernelGGL(k_planes, dim3(grid), dim3(256), 0, 0, r->d_crefcat, r->n_words, r->d_refplane + 2 * n_pairs, n_pairs);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return BSX_OK;
}

int bsx_index_build_wgbs(bsx_ref *r)
{
    const bsx_params &P = r->P;
    HIP_TRY(hipSetDevice(r->device));
    // visiting order: forward-copy blocks then rc-copy blocks (dbseq.cpp:441,468), each in sorted order
    std::vector<uint64_t> prefix(1, 0);
    std::vector<uint32_t> first, word0, anchor;
    std::vector<uint8_t> strand;
    for (int parity = 0; parity < 2; parity++)
        for (const Block &b : r->blocks) {
            if ((int)(b.id & 1) != parity) continue;
            const uint32_t I = P.index_interval, i0 = (b.begin / I) * I, i2 = ((b.end - P.seed_size) / I) * I;  // dbseq.cpp:352-353
            if (i2 < i0) continue;
            first.push_back(i0);
            word0.push_back(r->anchor[b.id >> 1] / BSX_SEGLEN);
            anchor.push_back(r->anchor[b.id >> 1]);
            strand.push_back((uint8_t)parity);
            prefix.push_back(prefix.back() + (i2 - i0) / I + 1);
        }
    const uint64_t total = prefix.back();
    if (total >= 0xFFFFFFFFull) { g_bsx_err = "index would exceed 2^32 entries"; return BSX_ERR_LIMIT; }
    const uint32_t K = P.total_kmers, nb = (uint32_t)first.size();
    DevBuf<uint32_t> d_off, d_nfwd, d_entries, d_keys, d_skeys, d_vals;
    DevBuf<uint64_t> d_prefix; DevBuf<uint32_t> d_first, d_word0, d_anchor; DevBuf<uint8_t> d_strand; DevBuf<char> d_temp;
    if (d_off.alloc((size_t)K + 1) || d_nfwd.alloc(K) || d_entries.alloc(total + BSX_ENTRY_PAD) || d_keys.alloc(total) || d_skeys.alloc(total) ||
        d_vals.alloc(total) || d_prefix.alloc(nb + 1) || d_first.alloc(nb) || d_word0.alloc(nb) || d_anchor.alloc(nb) || d_strand.alloc(nb)) {
        g_bsx_err = "hipMalloc failed while building the index";
        return BSX_ERR_NOMEM;
    }
    HIP_TRY(hipMemset(d_entries.p, 0, (total + BSX_ENTRY_PAD) * 4));
    if (total) {
        HIP_TRY(hipMemcpy(d_prefix.p, prefix.data(), (nb + 1) * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_first.p, first.data(), nb * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_word0.p, word0.data(), nb * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_anchor.p, anchor.data(), nb * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_strand.p, strand.data(), nb, hipMemcpyHostToDevice));
        BlockTab t{d_prefix.p, d_first.p, d_word0.p, d_anchor.p, d_strand.p, nb};
        const int grid = (int)std::min<uint64_t>((total + 255) / 256, 256 * 32);
        hipLaunchKernelGGL(k_enumerate, dim3(grid), dim3(256), 0, 0, t, total, r->d_refcat, r->d_crefcat, (uint32_t)P.seed_size,
                           (uint32_t)P.index_interval, P.seed_bits, d_keys.p, d_vals.p);
        HIP_TRY(hipGetLastError());
        unsigned bits = 1;
        while ((1ull << bits) < 2ull * K) bits++;
        size_t temp_bytes = 0;
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, d_keys.p, d_skeys.p, d_vals.p, d_entries.p, (size_t)total, 0u, bits, (hipStream_t)0));
        if (d_temp.alloc(temp_bytes)) { g_bsx_err = "hipMalloc failed for sort workspace"; return BSX_ERR_NOMEM; }
        HIP_TRY(rocprim::radix_sort_pairs(d_temp.p, temp_bytes, d_keys.p, d_skeys.p, d_vals.p, d_entries.p, (size_t)total, 0u, bits, (hipStream_t)0));
    }
    {
        const int grid = (int)std::min<uint64_t>(((uint64_t)K + 256) / 256, 256 * 32);
        hipLaunchKernelGGL(k_boundaries, dim3(grid), dim3(256), 0, 0, d_skeys.p, total, K, d_off.p, d_nfwd.p);
        HIP_TRY(hipGetLastError());
    }
    // the entries' context for the main kernel's prefilter (-I <= 4: its flank table holds four phases; BSX_CTX=0: none — 16 bytes per entry).  Not fatal:
    // without it the main kernel takes its plain path
    DevBuf<uint32_t> d_ctx;
    bool want_ctx = total && P.index_interval <= 4 && r->ctx_mode != 0 && !(getenv("BSX_CTX") && atoi(getenv("BSX_CTX")) == 0);   // (BSX_CTX=0: test hook)
    const uint64_t ctx_bytes = (total + BSX_ENTRY_PAD) * 16;
    if (want_ctx) {
        d_keys.reset(); d_vals.reset(); d_temp.reset();   // (the sort's inputs and workspace are spent)
        // The table buys speed, never results: it must not take the memory the batches need (a smaller or shared device, several lanes on one GPU).
        // Mode 1 (default) builds it only if `ctx_headroom` bytes stay free behind it — room for the fixed part of a 2^22-pair batch (20.6 GB of
        // per-wave slabs and per-unit arrays) plus the pools' reserve; a batch whose fixed part still does not fit drops the table and tries again
        // (bsx_batch_create).
        size_t fr = 0, tot = 0;
        if (r->ctx_mode == 1 && hipMemGetInfo(&fr, &tot) == hipSuccess && (uint64_t)fr < ctx_bytes + r->ctx_headroom) want_ctx = false;
    }
    if (want_ctx) {
        if (d_ctx.alloc((total + BSX_ENTRY_PAD) * 4) == 0) {
            HIP_TRY(hipMemset(d_ctx.p, 0, (total + BSX_ENTRY_PAD) * 16));
            const int grid = (int)std::min<uint64_t>((total + 255) / 256, 256 * 32);
            hipLaunchKernelGGL(k_context, dim3(grid), dim3(256), 0, 0, d_skeys.p, d_entries.p, total, r->d_refcat, r->d_crefcat, d_ctx.p);
            HIP_TRY(hipGetLastError());
        } else (void)hipGetLastError();
    }
    HIP_TRY(hipDeviceSynchronize());
    if (r->d_bucket_off) (void)hipFree(r->d_bucket_off);
    if (r->d_bucket_nfwd) (void)hipFree(r->d_bucket_nfwd);
    if (r->d_entries) (void)hipFree(r->d_entries);
    if (r->d_ctx) (void)hipFree(r->d_ctx);
    r->d_bucket_off = d_off.release(); r->d_bucket_nfwd = d_nfwd.release(); r->d_entries = d_entries.release(); r->d_ctx = d_ctx.release();
    r->ctx_bytes = r->d_ctx ? ctx_bytes : 0;
    r->n_entries = total;
    r->has_index = true;
    return BSX_OK;
}

// RRBS: the index only holds seeds anchored at digestion sites (dbseq.cpp:332-347,418-438) — 10^7 entries even for a
// human genome — so it is assembled on the host from the packed words and uploaded; entries are {tag, loc} pairs with
// tag = chr | seg<<16 | dir<<24 exactly as the reference's Hit.chr (dbseq.cpp:421,429).
int bsx_index_build_rrbs(bsx_ref *r, const std::vector<uint32_t> &refcat, const std::vector<uint32_t> &crefcat)
{
    const bsx_params &P = r->P;
    HIP_TRY(hipSetDevice(r->device));
    const uint32_t K = P.total_kmers, S = P.seed_size;
    // the kernels take a read's candidates from the (segment, direction) group of its bucket: 16 segments per direction (RRBS forces
    // seed 12, i.e. 12 segments of a 144-nt read; a shorter seed would need more groups than the table has)
    if (P.max_seedseg_num > 16) { g_bsx_err = "RRBS index: more than 16 seed segments per read"; return BSX_ERR_LIMIT; }
    if (r->d_ctx) { (void)hipFree(r->d_ctx); r->d_ctx = nullptr; r->ctx_bytes = 0; }   // (an RRBS index has no context table: the prefiltered kernel must never see a stale one)
    auto seed_at = [&](uint32_t chr, uint32_t loc) {
        const uint32_t *m = ((chr & 1) ? crefcat.data() : refcat.data()) + r->anchor[chr >> 1] / BSX_SEGLEN + (loc >> 4);
        const uint64_t v = ((uint64_t)m[0] << 32) | m[1];
        return bsx_seed_hash((uint32_t)(v >> (64 - 2 * S - 2 * (loc & 15))) & P.seed_bits);
    };
    std::vector<uint32_t> off((size_t)K + 1, 0), ent;
    const bool both = P.pairend || P.chains;
    for (int pass = 0; pass < 2; pass++) {
        std::vector<uint32_t> cur;
        if (pass) { cur.assign(off.begin(), off.end() - 1); }
        for (int j = 0; j < P.max_seedseg_num; j++)
            for (uint32_t chr = 0; chr < 2 * r->n_chr; chr++) {
                for (uint32_t loc : r->ccgg_index[j][chr]) {
                    uint32_t key = seed_at(chr, loc);
                    if (!pass) off[key + 1]++;
                    else { uint32_t q = cur[key]++; ent[2 * (size_t)q] = chr | ((uint32_t)j << 16); ent[2 * (size_t)q + 1] = loc; }
                }
                if (both) {
                    const uint32_t tmp_offset = r->rc_offset[chr >> 1] - S;
                    for (uint32_t it : r->ccgg_index[j][chr ^ 1]) {
                        uint32_t loc = tmp_offset - it, key = seed_at(chr, loc);
                        if (!pass) off[key + 1]++;
                        else { uint32_t q = cur[key]++; ent[2 * (size_t)q] = chr | ((uint32_t)j << 16) | 0x1000000u; ent[2 * (size_t)q + 1] = loc; }
                    }
                }
            }
        if (!pass) {
            for (uint32_t k = 0; k < K; k++) off[k + 1] += off[k];
            ent.assign(2 * (size_t)off[K] + 64, 0);
        }
    }
    // The kernels walk the entries of a bucket that carry the read's segment and direction (align.cpp:187,229).  The reference
    // filters the whole bucket; here every bucket is stably partitioned by (segment, direction) — the visiting order among
    // the entries of one group is unchanged, and that is all the reference's loop observes — with an offset table per
    // (bucket, group), so a read's candidates are one contiguous range (24 x fewer entries streamed at hg38 size).
    // bsx_index_download still returns the reference's own order (kept on the host).
    const bool grouped = true;
    std::vector<uint32_t> goff, gent;
    if (grouped) {
        goff.assign((size_t)K * 32 + 1, 0);
        gent.assign(ent.size(), 0);
        for (uint32_t k = 0; k < K; k++) {
            uint32_t cnt[32] = {0};
            for (uint32_t q = off[k]; q < off[k + 1]; q++) { const uint32_t t = ent[2 * (size_t)q]; cnt[((t >> 16) & 15u) + 16u * (t >> 24)]++; }
            uint32_t at = off[k], pos[32];
            for (int g = 0; g < 32; g++) { goff[(size_t)k * 32 + g] = at; pos[g] = at; at += cnt[g]; }
            for (uint32_t q = off[k]; q < off[k + 1]; q++) {
                const uint32_t t = ent[2 * (size_t)q], w = pos[((t >> 16) & 15u) + 16u * (t >> 24)]++;
                gent[2 * (size_t)w] = t; gent[2 * (size_t)w + 1] = ent[2 * (size_t)q + 1];
            }
        }
        goff[(size_t)K * 32] = off[K];
        r->rrbs_entries_host.assign(ent.begin(), ent.begin() + 2 * (size_t)off[K]);
    }
    std::vector<uint32_t> site_off(r->n_chr + 1, 0), sites_flat;
    for (uint32_t c = 0; c < r->n_chr; c++) { site_off[c + 1] = site_off[c] + (uint32_t)r->sites[c].size(); sites_flat.insert(sites_flat.end(), r->sites[c].begin(), r->sites[c].end()); }
    sites_flat.push_back(0);  // one-past-the-end read of the reference (dbseq.cpp:562) lands on a defined 0
    DevBuf<uint32_t> d_off, d_nfwd, d_ent, d_sites, d_soff;
    if (d_off.alloc((size_t)K + 1) || d_nfwd.alloc(K) || d_ent.alloc(ent.size()) || d_sites.alloc(sites_flat.size()) || d_soff.alloc(site_off.size()))
        return BSX_ERR_NOMEM;
    HIP_TRY(hipMemcpy(d_off.p, off.data(), ((size_t)K + 1) * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(d_nfwd.p, 0, (size_t)K * 4));
    HIP_TRY(hipMemcpy(d_ent.p, grouped ? gent.data() : ent.data(), ent.size() * 4, hipMemcpyHostToDevice));
    if (grouped) {
        DevBuf<uint32_t> d_goff;
        if (d_goff.alloc(goff.size())) return BSX_ERR_NOMEM;
        HIP_TRY(hipMemcpy(d_goff.p, goff.data(), goff.size() * 4, hipMemcpyHostToDevice));
        r->d_rrbs_goff = d_goff.release();
    }
    HIP_TRY(hipMemcpy(d_sites.p, sites_flat.data(), sites_flat.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_soff.p, site_off.data(), site_off.size() * 4, hipMemcpyHostToDevice));
    r->d_bucket_off = d_off.release(); r->d_bucket_nfwd = d_nfwd.release(); r->d_entries = d_ent.release();
    r->d_sites = d_sites.release(); r->d_site_off = d_soff.release();
    {   // where a position's 4 kb bin starts in its chromosome's site list: the fragment-size filter's search (ccgg_seglen) starts there
        std::vector<uint32_t> bin_off(r->n_chr + 1, 0), bins;
        for (uint32_t c = 0; c < r->n_chr; c++) {
            const std::vector<uint32_t> &sv = r->sites[c];
            const uint32_t top = std::max(r->chr_size[c], sv.empty() ? 0u : sv.back());
            const uint32_t nb = (top >> BSX_SITE_BIN_SHIFT) + 2;  // bins 0 .. top's bin, plus the end of the last one
            bin_off[c + 1] = bin_off[c] + nb;
            size_t i = 0;
            for (uint32_t b = 0; b < nb; b++) {
                const uint64_t lo = (uint64_t)b << BSX_SITE_BIN_SHIFT;
                while (i < sv.size() && sv[i] < lo) i++;
                bins.push_back((uint32_t)i);
            }
        }
        DevBuf<uint32_t> d_bin, d_boff;
        if (d_bin.alloc(bins.size()) || d_boff.alloc(bin_off.size())) return BSX_ERR_NOMEM;
        HIP_TRY(hipMemcpy(d_bin.p, bins.data(), bins.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_boff.p, bin_off.data(), bin_off.size() * 4, hipMemcpyHostToDevice));
        r->d_site_bin = d_bin.release(); r->d_site_bin_off = d_boff.release();
    }
    r->n_entries = off[K];
    r->has_index = true;
    return BSX_OK;
}
