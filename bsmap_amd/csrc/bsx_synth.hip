// bsx_synth.hip — deterministic synthetic workload generated ON the device (bench input).
//
// No reference counterpart: BSMAP reads FASTA/FASTQ files.  hg38 is not available on the GPU box, so bench.py
// measures on an "hg38-like" genome and bisulfite reads produced here, byte-reproducible from (seed, lengths):
//   genome  : per position a counter-based hash decides the base.  Background: GC 41 %, CpG observed/expected ~0.25
//             (a raw CpG keeps its G with probability 0.25, else G->A).  Every 512-nt window may carry one repeat
//             element: Alu-like (300 nt consensus, 12 % divergence, either orientation, ~18 % of windows = ~1.1 M
//             copies at hg38 size), L1-like (6144 nt consensus, fragments of 100-512 nt, 10 % divergence, ~8 % of
//             windows = ~0.5 M fragments), microsatellite ((TG)n,(CA)n,(A)n,(T)n,(TTTA)n,(GAA)n, 20-140 nt, 2 %
//             divergence, ~19 % of windows = ~3 % of bases).  N gaps: both telomeres (10 kb) and one centromere block
//             (4.4 % of the length at 40 %), i.e. ~5 % N in >= 10 kb runs.
//   reads   : uniform start over non-N sequence, Watson/Crick 50/50, bisulfite conversion (non-CpG C->T 99.5 %,
//             CpG C->T 25 %), 0.5 % substitutions; PE inserts ~N(300,50) clipped to [50,480], mate 2 is the reverse
//             complement of the fragment's far end, read-through is filled with adapter + random bases.
// The text form of a synthetic chromosome can be pulled back (bsx_synth_chr_text) so that tests can run the oracle's
// own packer/indexer on exactly the same sequence.
#include <algorithm>
#include <cstring>

#include "bsx_internal.h"

namespace {

typedef unsigned long long u64;

__host__ __device__ inline u64 mix64(u64 x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ inline u64 h3(u64 seed, u64 a, u64 b) { return mix64(mix64(seed ^ (a * 0xD1B54A32D192ED03ull)) ^ b); }
__host__ __device__ inline uint32_t u01_24(u64 h) { return (uint32_t)(h >> 40); }  // 24 uniform bits

// nucleotide index 0..3 = A C G T
__host__ __device__ inline int draw_base(uint32_t u24, uint32_t gc_permille)
{
    // P(C)=P(G)=gc/2, P(A)=P(T)=(1-gc)/2
    const uint32_t half_gc = (uint32_t)(((u64)gc_permille << 24) / 2000), half_at = (1u << 23) - half_gc;
    if (u24 < half_at) return 0;
    if (u24 < half_at + half_gc) return 1;
    if (u24 < half_at + 2 * half_gc) return 2;
    return 3;
}
__host__ __device__ inline int mutate_base(int b, u64 h) { return (b + 1 + (int)((h >> 20) % 3)) & 3; }

struct GapLayout { uint32_t tel, cen_begin, cen_end; };
__host__ __device__ inline GapLayout gaps_of(uint32_t len)
{
    GapLayout g;
    g.tel = len >= 400000 ? 10000u : len / 40;
    g.cen_begin = (uint32_t)((u64)len * 40 / 100);
    g.cen_end = g.cen_begin + (uint32_t)((u64)len * 44 / 1000);
    return g;
}

// base of chromosome c at position p: 0..3, or 4 for N
__host__ __device__ inline int synth_base(u64 seed, uint32_t c, uint32_t p, uint32_t len)
{
    const GapLayout g = gaps_of(len);
    if (p < g.tel || p >= len - g.tel || (p >= g.cen_begin && p < g.cen_end)) return 4;
    const uint32_t win = p >> 9, in = p & 511;
    const u64 wh = h3(seed, ((u64)c << 32) | win, 0x57494E);
    // bits 56-63 of the seed select a variant of the repeat content (bench.py's sensitivity leg): 0 as described above, 1 half the
    // microsatellite windows, 2 no repeat elements at all.  Everything else — and every genome made with a seed below 2^56 — is unchanged.
    const uint32_t variant = (uint32_t)(seed >> 56);
    const uint32_t sel = variant == 2 ? 1000u : (uint32_t)(wh % 1000);
    if (sel < 182) {  // Alu-like
        const uint32_t off = (uint32_t)((wh >> 12) % (512 - 300));
        if (in >= off && in < off + 300) {
            uint32_t k = in - off;
            const bool rc = (wh >> 40) & 1;
            if (rc) k = 299 - k;
            int b = draw_base(u01_24(h3(seed, 0xA1A1, k)), 560);
            if (rc) b = 3 - b;
            const u64 mh = h3(seed ^ wh, 0xD1, in);
            if (u01_24(mh) < (uint32_t)(0.12 * 16777216.0)) b = mutate_base(b, mh);
            return b;
        }
    } else if (sel < 265) {  // L1-like fragment
        const uint32_t flen = 100 + (uint32_t)((wh >> 12) % 413), off = (uint32_t)((wh >> 24) % (513 - flen));
        if (in >= off && in < off + flen) {
            const uint32_t cstart = (uint32_t)((wh >> 36) % (6144 - 512));
            uint32_t k = cstart + (in - off);
            const bool rc = (wh >> 50) & 1;
            if (rc) k = cstart + (flen - 1 - (in - off));
            int b = draw_base(u01_24(h3(seed, 0x1111, k)), 400);
            if (rc) b = 3 - b;
            const u64 mh = h3(seed ^ wh, 0xD2, in);
            if (u01_24(mh) < (uint32_t)(0.10 * 16777216.0)) b = mutate_base(b, mh);
            return b;
        }
    } else if (sel < (variant == 1 ? 360u : 455u)) {  // microsatellite
        const uint32_t mlen = 20 + (uint32_t)((wh >> 12) % 121), off = (uint32_t)((wh >> 24) % (513 - mlen));
        if (in >= off && in < off + mlen) {
            const uint32_t kind = (uint32_t)((wh >> 36) % 6), k = in - off;
            int b;
            switch (kind) {
            case 0: b = (k & 1) ? 2 : 3; break;                 // (TG)n
            case 1: b = (k & 1) ? 0 : 1; break;                 // (CA)n
            case 2: b = 0; break;                               // (A)n
            case 3: b = 3; break;                               // (T)n
            case 4: b = (k % 4 == 3) ? 0 : 3; break;            // (TTTA)n
            default: b = (k % 3 == 0) ? 2 : 0; break;           // (GAA)n
            }
            const u64 mh = h3(seed ^ wh, 0xD3, in);
            if (u01_24(mh) < (uint32_t)(0.02 * 16777216.0)) b = mutate_base(b, mh);
            return b;
        }
    }
    // background with CpG depletion
    const u64 bh = h3(seed, ((u64)c << 32) | p, 0xB6);
    int b = draw_base(u01_24(bh), 410);
    if (b == 2 && p > 0) {
        const int prev = draw_base(u01_24(h3(seed, ((u64)c << 32) | (p - 1), 0xB6)), 410);
        if (prev == 1 && ((bh >> 8) & 0xFFFF) >= 16384) b = 0;  // keep the G of a raw CpG with probability 1/4
    }
    return b;
}

__global__ void k_synth_pack(u64 seed, uint32_t c, uint32_t len, uint32_t n_words, uint32_t bit_nt_packed, uint32_t *__restrict__ fw,
                             uint32_t *__restrict__ rc)
{
    const uint32_t padded = n_words * 16;
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += gridDim.x * blockDim.x) {
        uint32_t x = 0, y = 0;
        for (uint32_t j = 0; j < 16; j++) {
            const uint32_t p = w * 16 + j, q = padded - 1 - p;  // q: forward position seen by the rc copy at index p
            const int bf = p < len ? synth_base(seed, c, p, len) : 4, br = q < len ? synth_base(seed, c, q, len) : 4;
            x = (x << 2) | ((bit_nt_packed >> (8 * (bf > 3 ? 0 : bf))) & 3u);       // alphabet[]: N -> code of A
            y = (y << 2) | ((bit_nt_packed >> (8 * (br > 3 ? 3 : 3 - br))) & 3u);   // rev_alphabet[]: N -> code of T
        }
        fw[w] = x; rc[w] = y;
    }
}

__global__ void k_synth_text(u64 seed, uint32_t c, uint32_t len, uint32_t from, uint32_t n, char *__restrict__ out)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = "ACGTN"[synth_base(seed, c, from + i, len)];
}

struct ReadGen {
    const uint32_t *refcat;
    const uint32_t *blk_gpos;    // [n] global nt coordinate of each forward block start
    const u64 *blk_prefix;       // [n+1] cumulative usable start positions
    uint32_t n_blk;
    uint32_t bit_nt_packed;
    u64 seed;
    uint32_t read_len, paired, first_unit;
    uint32_t kind;               // 0 plain, 1 trimming workload (low-quality 3' tails, adapter read-through), 2 RRBS (reads start at digestion sites)
    const uint32_t *sites, *site_off, *anchor;  // RRBS: digestion sites per chromosome (flat + offsets), chromosome anchors
    uint32_t n_chr, digest_len, digest_pos;
};

__device__ inline int ref_nt(const ReadGen &g, uint32_t gpos)
{
    const uint32_t code = (g.refcat[gpos >> 4] >> (30 - 2 * (gpos & 15))) & 3u;
    for (int i = 0; i < 4; i++) if (((g.bit_nt_packed >> (8 * i)) & 3u) == code) return i;
    return 0;
}

// one thread per unit: bench input generation is not on the timed path
__global__ void k_synth_reads(ReadGen g, uint32_t n, uint8_t *__restrict__ seq_a, uint8_t *__restrict__ seq_b, uint8_t *__restrict__ qual_a,
                              uint8_t *__restrict__ qual_b)
{
    const char ADAPTER[] = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA";
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += gridDim.x * blockDim.x) {
        const u64 id = g.first_unit + (u64)u;
        u64 r0 = h3(g.seed, id, 1);
        uint32_t ins = g.read_len;
        if (g.kind == 2) {
            // RRBS: the read starts at a digestion site; fragment = [site_k, site_k+1 + len(site) - 2 pos) as a
            // C-CGG digest leaves it, kept when it is read_len..220 nt long; Watson or Crick strand of the fragment
            const uint32_t n_sites = g.site_off[g.n_chr];
            uint32_t a = 0, b = 0;
            bool found = false;
            for (int tries = 0; tries < 64 && !found; tries++) {
                const uint32_t gi = (uint32_t)(h3(g.seed, id, 100 + tries) % (u64)n_sites);
                uint32_t lo = 0, hi = g.n_chr;
                while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (g.site_off[mid] <= gi) lo = mid; else hi = mid; }
                if (gi + 1 >= g.site_off[lo + 1]) continue;
                const uint32_t fa = g.sites[gi], fb = g.sites[gi + 1] + g.digest_len - 2 * g.digest_pos;
                if (fb - fa < g.read_len || fb - fa > 220) continue;
                a = g.anchor[lo] + fa; b = g.anchor[lo] + fb; found = true;
            }
            if (!found) { a = g.anchor[0] + g.sites[0]; b = a + g.read_len; }  // (a genome without a usable fragment: any site)
            const bool watson = (r0 >> 63) == 0;
            uint8_t *out = seq_a + (size_t)u * g.read_len;
            for (uint32_t k = 0; k < g.read_len; k++) {
                int base, next;
                if (watson) { base = ref_nt(g, a + k); next = ref_nt(g, a + k + 1); }
                else { base = 3 - ref_nt(g, b - 1 - k); next = b - 2 - k >= a ? 3 - ref_nt(g, b - 2 - k) : 0; }
                if (base == 1) {
                    const uint32_t uc = u01_24(h3(g.seed ^ 0xB15, id, k));
                    if (uc < (next == 2 ? (uint32_t)(0.25 * 16777216.0) : (uint32_t)(0.995 * 16777216.0))) base = 3;
                }
                const u64 rk = h3(g.seed ^ 0x5EED, id * 2, k);
                if (u01_24(rk) < (uint32_t)(0.005 * 16777216.0)) base = mutate_base(base, rk);
                out[k] = "ACGT"[base];
                if (qual_a) qual_a[(size_t)u * g.read_len + k] = 'I';
            }
            continue;
        }
        if (g.paired) {  // ~N(300,50) from 4 uniforms, clipped
            const u64 r1 = h3(g.seed, id, 2);
            const int s4 = (int)(r1 & 0xFFFF) + (int)((r1 >> 16) & 0xFFFF) + (int)((r1 >> 32) & 0xFFFF) + (int)((r1 >> 48) & 0xFFFF);
            int v = 300 + (int)(((long long)(s4 - 131070) * 50) / 37837);  // sd of the 4-uniform sum = 37837
            v = v < 50 ? 50 : (v > 480 ? 480 : v);
            ins = (uint32_t)v;
        }
        if (g.kind == 1) {  // trimming workload: 30 % of the fragments are 30-149 nt long, so the reads run into the adapter
            const u64 r2 = h3(g.seed, id, 3);
            if (u01_24(r2) < (uint32_t)(0.30 * 16777216.0)) ins = 30 + (uint32_t)((r2 & 0xffffff) % 120);
        }
        // uniform fragment start over positions where the whole fragment stays inside a block
        const u64 total = g.blk_prefix[g.n_blk];
        u64 t = r0 % total;
        uint32_t lo = 0, hi = g.n_blk;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (g.blk_prefix[mid] <= t) lo = mid; else hi = mid; }
        const uint32_t start = g.blk_gpos[lo] + (uint32_t)(t - g.blk_prefix[lo]);
        const bool watson = (r0 >> 63) == 0;
        // mate m base k: converted strand s[k] for m=0, revcomp(s)[k] for m=1
        const int nm = g.paired ? 2 : 1;
        for (int m = 0; m < nm; m++) {
            uint8_t *out = (m ? seq_b : seq_a) + (size_t)u * g.read_len;
            uint8_t *qout = (m ? qual_b : qual_a) ? (m ? qual_b : qual_a) + (size_t)u * g.read_len : nullptr;
            const uint32_t tail = g.kind == 1 ? 10 + (uint32_t)(h3(g.seed, id * 2 + m, 4) % 51) : 0;  // 3' tail of 10-60 nt with Q2-Q15
            for (uint32_t k = 0; k < g.read_len; k++) {
                int b;
                if (qout) qout[k] = (k + tail >= g.read_len) ? (uint8_t)(33 + 2 + (h3(g.seed ^ 0x9A1, id * 2 + m, k) % 14)) : (uint8_t)'I';
                const u64 rk = h3(g.seed ^ 0x5EED, id * 2 + m, k);
                const bool inside = m == 0 ? k < ins : k < ins;
                if (inside) {
                    // index on the converted strand, 0 = 5' end
                    const uint32_t sidx = m == 0 ? k : ins - 1 - k;
                    // strand base and its 3' neighbour, in forward-genome terms
                    int base, next;
                    if (watson) { base = ref_nt(g, start + sidx); next = sidx + 1 < ins + 1 ? ref_nt(g, start + sidx + 1) : 0; }
                    else { base = 3 - ref_nt(g, start + ins - 1 - sidx); next = (start + ins - 1 - sidx) > 0 ? 3 - ref_nt(g, start + ins - 2 - sidx) : 0; }
                    // bisulfite conversion of C on this strand; the decision is a function of the strand position only,
                    // so both mates see the same converted molecule
                    if (base == 1) {
                        const uint32_t uc = u01_24(h3(g.seed ^ 0xB15, id, sidx));
                        const uint32_t pconv = next == 2 ? (uint32_t)(0.25 * 16777216.0) : (uint32_t)(0.995 * 16777216.0);
                        if (uc < pconv) base = 3;
                    }
                    b = m == 0 ? base : 3 - base;
                } else {
                    const uint32_t a = k - ins;
                    b = a < sizeof(ADAPTER) - 1 ? (ADAPTER[a] == 'A' ? 0 : ADAPTER[a] == 'C' ? 1 : ADAPTER[a] == 'G' ? 2 : 3) : (int)(rk & 3);
                }
                if (u01_24(rk) < (uint32_t)(0.005 * 16777216.0)) b = mutate_base(b, rk);  // sequencing error
                out[k] = "ACGT"[b];
            }
        }
    }
}

}  // namespace

static void synth_blocks(uint32_t c, uint32_t len, uint32_t padded, std::vector<Block> &blocks)
{
    const GapLayout g = gaps_of(len);
    const uint32_t seg[2][2] = {{g.tel, g.cen_begin}, {g.cen_end, len - g.tel}};
    for (auto &s : seg) {
        if (s[1] <= s[0] || s[1] - s[0] < 30) continue;  // UnmaskRegion keeps runs >= 30 nt (dbseq.cpp:127)
        blocks.push_back(Block{2 * c, s[0], s[1]});
        blocks.push_back(Block{2 * c + 1, padded - s[1], padded - s[0]});
    }
}

extern "C" int bsx_ref_create_synthetic(const bsx_params *p, uint32_t n_chr, const uint32_t *chr_len, uint64_t seed, int device, bsx_ref **out)
{
    if (!p || !chr_len || !out || n_chr == 0) return BSX_ERR_ARG;
    if (p->rrbs) {
        // RRBS mode needs the digestion-site tables and the site-anchored index, which the host packer builds from the
        // text: generate the same genome as text (on the device), hand it to the ordinary loader as a FASTA image
        bsx_params q = *p;
        q.rrbs = 0;
        bsx_ref *tmp = nullptr;
        int rc = bsx_ref_create_synthetic(&q, n_chr, chr_len, seed, device, &tmp);
        if (rc) return rc;
        std::string fa;
        uint64_t total = 0;
        for (uint32_t c = 0; c < n_chr; c++) total += chr_len[c] + 16;
        fa.reserve(total);
        std::vector<char> buf;
        for (uint32_t c = 0; c < n_chr && rc == BSX_OK; c++) {
            fa += ">" + tmp->names[c] + "\n";
            buf.resize(chr_len[c]);
            rc = bsx_synth_chr_text(tmp, c, 0, chr_len[c], buf.data());
            fa.append(buf.data(), chr_len[c]);
            fa += "\n";
        }
        bsx_ref_destroy(tmp);
        if (rc) return rc;
        rc = bsx_ref_create_from_fasta(p, fa.data(), fa.size(), device, out);
        if (rc == BSX_OK) (*out)->synth_seed = seed;
        return rc;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device >= ndev) { g_bsx_err = "no HIP device"; return BSX_ERR_NODEVICE; }
    HIP_TRY(hipSetDevice(device));
    bsx_ref *r = new bsx_ref();
    r->P = *p; r->device = device; r->n_chr = n_chr;
    uint64_t words = 0;
    r->anchor.assign(1, BSX_REF_MARGIN * BSX_SEGLEN);
    for (uint32_t c = 0; c < n_chr; c++) {
        if (chr_len[c] < 64) { delete r; return BSX_ERR_ARG; }
        const uint32_t nw = (chr_len[c] + 15) / 16 + 2;
        char nm[32]; snprintf(nm, sizeof(nm), "syn%u", c + 1);
        r->names.push_back(nm);
        r->chr_size.push_back(chr_len[c]);
        r->rc_offset.push_back(nw * 16);
        synth_blocks(c, chr_len[c], nw * 16, r->blocks);
        words += nw;
        if ((words + 2 * BSX_REF_MARGIN) * 16 >= 0xFFFFFFFFull) { delete r; return BSX_ERR_LIMIT; }
        r->anchor.push_back((uint32_t)((words + BSX_REF_MARGIN) * 16));
        r->sum_length += chr_len[c];
    }
    std::sort(r->blocks.begin(), r->blocks.end(), [](const Block &a, const Block &b) { return a.id < b.id || (a.id == b.id && a.begin < b.begin); });
    r->n_words = words + 2 * BSX_REF_MARGIN;
    auto fail = [&](int rc) { bsx_ref_destroy(r); return rc; };
    if (hipMalloc((void **)&r->d_refcat, 2 * (r->n_words + 64) * 4) != hipSuccess) return fail(BSX_ERR_NOMEM);  // one allocation, see finish_ref_upload
    r->d_crefcat = r->d_refcat + r->n_words + 64;
    if (hipMemset(r->d_refcat, 0, 2 * (r->n_words + 64) * 4) != hipSuccess) return fail(BSX_ERR_DEVICE);
    const uint32_t bnp = p->bit_nt[0] | (p->bit_nt[1] << 8) | (p->bit_nt[2] << 16) | ((uint32_t)p->bit_nt[3] << 24);
    for (uint32_t c = 0; c < n_chr; c++) {
        const uint32_t nw = r->rc_offset[c] / 16, w0 = r->anchor[c] / 16;
        const int grid = (int)std::min<uint32_t>((nw + 255) / 256, 256 * 16);
        hipLaunchKernelGGL(k_synth_pack, dim3(grid), dim3(256), 0, 0, (u64)seed, c, chr_len[c], nw, bnp, r->d_refcat + w0, r->d_crefcat + w0);
    }
    if (hipDeviceSynchronize() != hipSuccess) return fail(BSX_ERR_DEVICE);
    { const int prc = bsx_planes_build(r); if (prc != BSX_OK) return fail(prc); }
    if (hipMalloc((void **)&r->d_anchor, r->anchor.size() * 4) != hipSuccess || hipMalloc((void **)&r->d_chr_size, n_chr * 4) != hipSuccess ||
        hipMalloc((void **)&r->d_rc_offset, n_chr * 4) != hipSuccess) return fail(BSX_ERR_NOMEM);
    (void)hipMemcpy(r->d_anchor, r->anchor.data(), r->anchor.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(r->d_chr_size, r->chr_size.data(), n_chr * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(r->d_rc_offset, r->rc_offset.data(), n_chr * 4, hipMemcpyHostToDevice);
    r->synth_seed = seed;
    *out = r;
    return BSX_OK;
}

extern "C" int bsx_synth_chr_text(const bsx_ref *r, uint32_t c, uint32_t from, uint32_t n, char *out)
{
    if (!r || c >= r->n_chr || !out || (uint64_t)from + n > r->chr_size[c]) return BSX_ERR_ARG;
    HIP_TRY(hipSetDevice(r->device));
    char *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, n ? n : 1));
    hipLaunchKernelGGL(k_synth_text, dim3(std::min<uint32_t>((n + 255) / 256, 4096)), dim3(256), 0, 0, (u64)r->synth_seed, c, r->chr_size[c], from, n, d);
    hipError_t e = hipMemcpy(out, d, n, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    HIP_TRY(e);
    return BSX_OK;
}

int bsx_synth_reads_launch(const bsx_ref *r, uint32_t n, uint32_t read_len, int paired, uint64_t seed, uint32_t first_index, uint8_t *d_seq_a,
                           uint8_t *d_seq_b, hipStream_t stream, int kind, uint8_t *d_qual_a, uint8_t *d_qual_b)
{
    if (kind == 2 && (!r->P.rrbs || !r->d_sites || paired)) { g_bsx_err = "RRBS reads need an RRBS reference and a single-end batch"; return BSX_ERR_ARG; }
    // usable fragment starts: positions of forward blocks where a fragment of the maximum span still fits
    const uint32_t span = paired ? 481 : read_len + 1;
    std::vector<uint32_t> gpos; std::vector<u64> prefix(1, 0);
    for (const Block &b : r->blocks) {
        if (b.id & 1) continue;
        if (b.end - b.begin <= span) continue;
        gpos.push_back(r->anchor[b.id >> 1] + b.begin);
        prefix.push_back(prefix.back() + (b.end - b.begin - span));
    }
    if (gpos.empty()) { g_bsx_err = "reference has no block long enough to sample reads from"; return BSX_ERR_STATE; }
    uint32_t *d_gpos = nullptr; u64 *d_prefix = nullptr;
    HIP_TRY(hipMalloc((void **)&d_gpos, gpos.size() * 4));
    HIP_TRY(hipMalloc((void **)&d_prefix, prefix.size() * 8));
    HIP_TRY(hipMemcpy(d_gpos, gpos.data(), gpos.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_prefix, prefix.data(), prefix.size() * 8, hipMemcpyHostToDevice));
    ReadGen g;
    g.refcat = r->d_refcat; g.blk_gpos = d_gpos; g.blk_prefix = d_prefix; g.n_blk = (uint32_t)gpos.size();
    g.bit_nt_packed = r->P.bit_nt[0] | (r->P.bit_nt[1] << 8) | (r->P.bit_nt[2] << 16) | ((uint32_t)r->P.bit_nt[3] << 24);
    g.seed = seed; g.read_len = read_len; g.paired = paired; g.first_unit = first_index;
    g.kind = (uint32_t)kind; g.sites = r->d_sites; g.site_off = r->d_site_off; g.anchor = r->d_anchor; g.n_chr = r->n_chr;
    g.digest_len = (uint32_t)strlen(r->P.digest_site); g.digest_pos = (uint32_t)r->P.digest_pos;
    hipLaunchKernelGGL(k_synth_reads, dim3(std::min<uint32_t>((n + 255) / 256, 256 * 16)), dim3(256), 0, stream, g, n, d_seq_a, d_seq_b, d_qual_a, d_qual_b);
    hipError_t e = hipStreamSynchronize(stream);
    (void)hipFree(d_gpos); (void)hipFree(d_prefix);
    HIP_TRY(e);
    return BSX_OK;
}
