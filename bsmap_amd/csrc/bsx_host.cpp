// bsx_host.cpp — host side of libbsx.so that needs no device code: option semantics, alphabet,
// seed profile and the reference packer.
//
// Reference behaviour restated here (file:line in BSMAP v2.6):
//   Param::Param / SetSeedSize / SetDigestionSite / SetAlign / InitMapping   param.cpp:6-121,187-231
//   RefSeq::LoadNextSeq / BinSeq / cBinSeq / UnmaskRegion / Run_ConvertBinseq dbseq.cpp:18-142,215-282
//   RefSeq::find_CCGG                                                          dbseq.cpp:144-211
#include <algorithm>
#include <atomic>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <thread>

#include "bsx_cpus.h"
#include "bsx_internal.h"

thread_local std::string g_bsx_err;

int bsx_hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    char buf[512];
    snprintf(buf, sizeof(buf), "%s:%d: %s -> %s", file, line, what, hipGetErrorString(e));
    g_bsx_err = buf;
    if (e == hipErrorOutOfMemory) return BSX_ERR_NOMEM;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return BSX_ERR_NODEVICE;
    return BSX_ERR_DEVICE;
}

extern "C" const char *bsx_last_error_detail(void) { return g_bsx_err.c_str(); }

extern "C" const char *bsx_strerror(int code)
{
    switch (code) {
    case BSX_OK: return "ok";
    case BSX_ERR_ARG: return "invalid argument";
    case BSX_ERR_IO: return "cannot read input";
    case BSX_ERR_NOMEM: return "out of memory";
    case BSX_ERR_DEVICE: return "HIP runtime error";
    case BSX_ERR_STATE: return "call order violated";
    case BSX_ERR_LIMIT: return "reference limit exceeded";
    case BSX_ERR_NODEVICE: return "no gfx950 device available (libbsx has no CPU fallback)";
    }
    return "unknown error";
}

extern "C" int bsx_params_default(bsx_params *p)
{
    if (!p) return BSX_ERR_ARG;
    memset(p, 0, sizeof(*p));
    p->seed_size = 16;              // param.cpp:44
    p->index_interval = 4;          // param.cpp:76
    p->max_snp_num = 2;             // param.cpp:49
    p->max_num_hits = BSX_MAXHITS;  // param.cpp:50
    p->min_insert = 28;             // param.cpp:40
    p->max_insert = 500;            // param.cpp:41
    p->report_repeat_hits = 1;      // param.cpp:56
    p->zero_qual = '!';             // param.cpp:36
    p->max_ns = 5;                  // param.cpp:33
    p->max_readlen = BSX_MAX_READLEN;
    p->read_nt = 'T';
    p->ref_nt = 'C';                // param.cpp:81
    return BSX_OK;
}

extern "C" int bsx_params_set_digest(bsx_params *p, const char *site)
{
    if (!p || !site) return BSX_ERR_ARG;
    const char *dash = strchr(site, '-');
    if (!dash) return BSX_ERR_ARG;  // "Digestion position not marked" (param.cpp:98-101)
    size_t n = strlen(site);
    if (n < 2 || n - 1 >= 16) return BSX_ERR_ARG;
    p->digest_pos = (int)(dash - site);
    memset(p->digest_site, 0, sizeof(p->digest_site));
    memcpy(p->digest_site, site, p->digest_pos);
    strcpy(p->digest_site + p->digest_pos, dash + 1);
    p->rrbs = 1;
    p->index_interval = 1;
    p->seed_size = 12;
    return BSX_OK;
}

static int nt_index(int c)
{
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; }
    return -1;
}

extern "C" int bsx_params_finish(bsx_params *p)
{
    if (!p) return BSX_ERR_ARG;
    if (p->rrbs) { p->seed_size = 12; p->index_interval = 1; }  // main.cpp:247,257
    if (p->seed_size < 1 || p->seed_size > 16) return BSX_ERR_LIMIT;
    if (p->index_interval < 1 || p->index_interval > 16) return BSX_ERR_LIMIT;      // main.cpp:258
    if (p->max_snp_num < 0 || p->max_snp_num > BSX_MAXSNPS) return BSX_ERR_LIMIT;    // main.cpp:260-262
    if (p->max_num_hits < 1 || p->max_num_hits > BSX_MAXHITS) return BSX_ERR_LIMIT;  // main.cpp:263-265
    if (p->max_readlen < 1 || p->max_readlen > BSX_MAX_READLEN) p->max_readlen = BSX_MAX_READLEN;
    if (p->n_adapter < 0 || p->n_adapter > 10) return BSX_ERR_ARG;
    int rd = nt_index(p->read_nt), rf = nt_index(p->ref_nt);
    if (rd < 0 || rf < 0) return BSX_ERR_ARG;  // "Unknown nucleotide."
    if (rd == rf) return BSX_ERR_ARG;          // "Must specify different nucleotides for additional alignment."
    // SetAlign: the read nucleotide gets code 3, the reference nucleotide code 1, the other two 0 and 2 in ACGT order
    uint8_t other = 0;
    for (int i = 0; i < 4; i++) {
        if (i == rd) p->bit_nt[i] = 3;
        else if (i == rf) p->bit_nt[i] = 1;
        else { p->bit_nt[i] = other; other = 2; }
    }
    p->seed_bits = 0;
    for (int i = 0; i < p->seed_size; i++) p->seed_bits |= 3u << (2 * i);
    memset(p->profile_a, 0, sizeof(p->profile_a));
    for (int ph = 0; ph < p->index_interval; ph++)
        for (int seg = 0; seg <= BSX_MAXSNPS; seg++) {
            int a = ((seg * p->seed_size + ph + p->index_interval - 1) / p->index_interval) * p->index_interval;
            p->profile_a[seg][ph] = (uint8_t)a;  // bit8_t in the reference: wraps the same way
        }
    p->max_seedseg_num = 9 * 16 / p->seed_size;  // dbseq.cpp:217
    p->total_kmers = 1;
    for (int i = 0; i < p->seed_size; i++) p->total_kmers *= 3;
    return BSX_OK;
}

// ---------------------------------------------------------------------------------------------
// reference packer
// ---------------------------------------------------------------------------------------------
namespace {

inline bool is_space(char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

struct Cursor {
    const char *t; uint64_t n, i;
    void skip_ws() { while (i < n && is_space(t[i])) i++; }
    bool token(uint64_t &b, uint64_t &e) { skip_ws(); b = i; while (i < n && !is_space(t[i])) i++; e = i; return e > b; }
};

// one FASTA record: its name and the text region holding the sequence tokens
struct Rec { std::string name; uint64_t sb, se; };

// everything derived from one record; records are independent, so they are packed by a pool of threads
struct Packed {
    int rc = BSX_OK;
    uint32_t L = 0, padded = 0;
    std::vector<uint32_t> f, c;
    std::vector<Block> blocks;
    std::vector<uint32_t> sites;
    std::vector<std::vector<uint32_t>> ccgg_f, ccgg_r;  // [segment]
};

// chunk sizes of the parallel packer (tests/harness/pack_check.cpp builds with tiny ones so that a 100 kb FASTA crosses hundreds of chunk borders)
#ifndef BSX_PACK_CHUNK
#define BSX_PACK_CHUNK (4u << 20)       /* text bytes per task of the count / copy phases */
#define BSX_PACK_WORDS (1u << 18)       /* packed words per task */
#define BSX_PACK_NXCHUNK (8u << 20)     /* characters per task of the N/X run listing */
#define BSX_PACK_BIG (32u << 20)        /* records from this size on are packed chunk-parallel, one after another */
#define BSX_PACK_MINTEXT (8u << 20)     /* texts below this: one thread */
#endif
// run fn(k) for k in [0, n) on up to nt threads (the calling thread included); n small: chunks of a record
template <class F> void pfor(size_t n, unsigned nt, F fn)
{
    if (nt <= 1 || n <= 1) { for (size_t k = 0; k < n; k++) fn(k); return; }
    std::atomic<size_t> next(0);
    auto work = [&] { for (size_t k; (k = next.fetch_add(1)) < n;) fn(k); };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < std::min<size_t>(nt, n); t++) th.emplace_back(work);
    work();
    for (std::thread &t : th) t.join();
}

// One record -> packed words of both strand copies, unmasked blocks, RRBS site tables.  nt threads work on CHUNKS of the record (round 6): a human chromosome is
// 250 MB of text and one thread per record left the whole load waiting for chromosome 1 (2.25 s of a 6.9 s run at hg38 size; 24 records on 16 CPUs).  Every
// phase streams the record once: count the non-blank characters per chunk, copy them to their place, pack word ranges, list the N/X runs per chunk.
void pack_record(const bsx_params &P, const char *text, const Rec &R, uint32_t chr, const uint8_t *code_f, const uint8_t *code_r,
                 const uint8_t *cls, Packed &o, unsigned nt = 1)
{
    // concatenate the whitespace-separated tokens (dbseq.cpp:40-50)
    const uint64_t span = R.se - R.sb, CH = BSX_PACK_CHUNK;
    const size_t nck = (size_t)std::max<uint64_t>(1, (span + CH - 1) / CH);
    std::vector<uint64_t> cnt(nck + 1, 0);
    pfor(nck, nt, [&](size_t k) {
        const uint64_t a = R.sb + k * CH, b = std::min(R.se, a + CH);
        uint64_t c = 0;
        for (uint64_t i = a; i < b; i++) c += !(cls[(uint8_t)text[i]] & 4);
        cnt[k + 1] = c;
    });
    for (size_t k = 0; k < nck; k++) cnt[k + 1] += cnt[k];
    const uint64_t len = cnt[nck];
    if (len >= 0xFFFFFFFFull - 64) { o.rc = BSX_ERR_LIMIT; return; }
    const uint32_t L = (uint32_t)len;
    const uint32_t nw = (L + BSX_SEGLEN - 1) / BSX_SEGLEN + 2;  // BinSeq: two spare words (dbseq.cpp:60)
    const uint64_t padded = (uint64_t)nw * BSX_SEGLEN;
    std::unique_ptr<char[]> bufp(new char[padded + 64]);   // (not value-initialised: every byte below `padded` is written by the two loops that follow)
    char *const buf = bufp.get();
    const uint64_t buf_size = padded + 64;
    pfor(nck, nt, [&](size_t k) {
        const uint64_t a = R.sb + k * CH, b = std::min(R.se, a + CH);
        uint64_t q = cnt[k];
        const uint64_t lim = cnt[k + 1];   // (chunks write side by side: never past the chunk's own last character)
        for (uint64_t i = a; i < b && q < lim; i++) { const char ch = text[i]; buf[q] = ch; q += !(cls[(uint8_t)ch] & 4); }
    });
    std::fill(buf + L, buf + buf_size, 'N');
    o.L = L; o.padded = (uint32_t)padded;
    o.f.resize(nw); o.c.resize(nw);
    const uint32_t WCH = BSX_PACK_WORDS;   // words per task (4 M nt)
    pfor((nw + WCH - 1) / WCH, nt, [&](size_t k) {
        const uint32_t w0 = (uint32_t)k * WCH, w1 = std::min(nw, w0 + WCH);
        for (uint32_t w = w0; w < w1; w++) {
            uint32_t x = 0, y = 0;
            const uint64_t base = (uint64_t)w * BSX_SEGLEN, rbase = padded - 1 - base;
            for (uint32_t j = 0; j < BSX_SEGLEN; j++) {
                x = (x << 2) | code_f[(uint8_t)buf[base + j]];
                y = (y << 2) | code_r[(uint8_t)buf[rbase - j]];
            }
            o.f[w] = x; o.c[w] = y;
        }
    });
    // UnmaskRegion (dbseq.cpp:114-142): maximal runs between N/X characters that start at an ACGT letter and are
    // >= 30 nt; each run is recorded on the forward copy (id 2c) and mirrored on the rc copy (id 2c+1).  The
    // reference scans a work string that still holds older records behind this one; the 'N' padding written above
    // stops both scans before that tail, so the record alone decides.
    // The reference's two alternating scans — to the next ACGT letter, then to the next N/X — only ever stop at the ends of N/X runs or inside the stretches
    // between them: the N/X runs of [0, padded) are listed per chunk in parallel (a handful per chromosome), joined where they touch, and the scans then jump
    // over them.
    {
        const uint64_t NCH = BSX_PACK_NXCHUNK;
        const size_t nnk = (size_t)((padded + NCH - 1) / NCH);
        std::vector<std::vector<std::pair<uint64_t, uint64_t>>> runs_k(nnk);
        pfor(nnk, nt, [&](size_t k) {
            const uint64_t a = k * NCH, b = std::min<uint64_t>(padded, a + NCH);
            uint64_t i = a;
            while (i < b) {
                if (!(cls[(uint8_t)buf[i]] & 2)) { i++; continue; }
                uint64_t j = i + 1;
                while (j < b && (cls[(uint8_t)buf[j]] & 2)) j++;
                runs_k[k].emplace_back(i, j);
                i = j;
            }
        });
        std::vector<std::pair<uint64_t, uint64_t>> runs;   // maximal N/X runs [start, end) of the padded record, ascending
        for (auto &v : runs_k) for (auto &r_ : v) { if (!runs.empty() && runs.back().second == r_.first) runs.back().second = r_.second; else runs.push_back(r_); }
        size_t ri = 0;   // first run that ends behind the scan position
        uint32_t begin, end = 0;
        while (end < L) {
            uint64_t q = end;
            // while (q < padded && !(cls & 1)) q++   — N/X runs are skipped whole
            for (;;) {
                while (ri < runs.size() && runs[ri].second <= q) ri++;
                if (ri < runs.size() && runs[ri].first <= q) { q = runs[ri].second; continue; }
                if (q >= padded || (cls[(uint8_t)buf[q]] & 1)) break;
                q++;
            }
            if (q >= padded || q > L) break;
            begin = (uint32_t)q;
            // while (q < padded && !(cls & 2)) q++   — the next N/X character is the start of the next run
            while (ri < runs.size() && runs[ri].second <= q) ri++;
            q = ri < runs.size() ? std::max<uint64_t>(q, runs[ri].first) : padded;
            end = q <= L ? (uint32_t)q : L;
            if (end - begin < 30) continue;
            // (the reference's "merge with previous block if gap < 5" test compares against the rc twin pushed
            //  last and therefore never fires)
            o.blocks.push_back(Block{2 * chr, begin, end});
            o.blocks.push_back(Block{2 * chr + 1, (uint32_t)padded - end, (uint32_t)padded - begin});
        }
    }
    if (P.rrbs) {
        // find_CCGG (dbseq.cpp:144-211)
        const size_t dl = strlen(P.digest_site);
        const uint64_t bs = buf_size;
        for (uint64_t q = 0; q < L && q + dl <= bs; q++) {
            bool ok = true;
            for (size_t k = 0; k < dl && ok; k++) ok = toupper((uint8_t)buf[q + k]) == P.digest_site[k];
            if (ok) o.sites.push_back((uint32_t)q + P.digest_pos);
        }
        // Seed positions of the restriction fragments (the rest of find_CCGG).  A fragment between neighbouring cuts no further
        // apart than -x is indexed from both of its ends: a read of the forward strand starts at the fragment's left cut, so its
        // segment g lies at cut + g S on the forward copy; a read of the other strand starts at the right end of the fragment
        // (right cut + site length - 2 cut offsets), which on the rc copy is position padded - end, and its segment g lies g S
        // behind that.  Each segment's two lists are filled in ascending site order — CCGG_index[g][2c] / [2c + 1].
        const uint32_t S = (uint32_t)P.seed_size, span = (uint32_t)dl - 2u * (uint32_t)P.digest_pos, reach = (uint32_t)P.max_insert;
        const uint32_t last_start = L - S;   // (32-bit arithmetic as in the reference: wraps for a record shorter than one seed)
        const std::vector<uint32_t> &cuts = o.sites;
        const size_t nc = cuts.size();
        o.ccgg_f.assign(P.max_seedseg_num, {}); o.ccgg_r.assign(P.max_seedseg_num, {});
        for (int g = 0; g < P.max_seedseg_num; g++) {
            std::vector<uint32_t> &fwd = o.ccgg_f[g], &rev = o.ccgg_r[g];
            const uint32_t shift = (uint32_t)g * S;
            for (size_t k = 0; k < nc; k++) {
                const bool is_left_cut = k + 1 < nc && cuts[k + 1] - cuts[k] <= reach, is_right_cut = k > 0 && cuts[k] - cuts[k - 1] <= reach;
                if (is_left_cut && cuts[k] + shift <= last_start) fwd.push_back(cuts[k] + shift);
                const uint32_t end = cuts[k] + span;   // one past the fragment on the forward copy
                if (is_right_cut && (uint64_t)end >= (uint64_t)shift + S) rev.push_back((uint32_t)padded - end + shift);
            }
        }
    }
}

}  // namespace

int bsx_pack_fasta(const bsx_params &P, const char *text, uint64_t n, bsx_ref &r, std::vector<uint32_t> &refcat,
                   std::vector<uint32_t> &crefcat)
{
    // The reference reads records with operator>> (dbseq.cpp:18-54): the first non-blank character is consumed
    // unchecked, the next token is the name, the rest of that line is ignored, then whitespace-separated tokens
    // are concatenated until one begins with '>'.  Records are located in one sequential pass and packed in parallel.
    uint8_t code_f[256], code_r[256], cls[256];  // cls: 1 = Param::useful_nt, 2 = Param::nx_nt, 4 = whitespace
    for (int c = 0; c < 256; c++) {
        int k = nt_index(c);
        code_f[c] = P.bit_nt[k < 0 ? 0 : k];          // alphabet[]: unknown -> code of 'A'
        code_r[c] = P.bit_nt[k < 0 ? 3 : 3 - k];      // rev_alphabet[]: unknown -> code of 'T'
        cls[c] = (uint8_t)((k >= 0 ? 1 : 0) | ((c == 'N' || c == 'X' || c == 'n' || c == 'x') ? 2 : 0) | (is_space((char)c) ? 4 : 0));
    }
    Cursor cur{text, n, 0};
    std::vector<Rec> recs;
    for (;;) {
        cur.skip_ws();
        if (cur.i >= cur.n) break;
        cur.i++;  // fin>>c
        uint64_t b, e;
        cur.token(b, e);
        Rec R;
        R.name.assign(text + b, text + e);
        const char *nl = (const char *)memchr(text + cur.i, '\n', cur.n - cur.i);
        cur.i = nl ? (uint64_t)(nl - text) + 1 : cur.n;
        cur.skip_ws();
        if (cur.i >= cur.n || text[cur.i] == '>') break;  // empty sequence: LoadNextSeq returned 0, loading stops
        R.sb = cur.i;
        // the record ends in front of the next token that starts with '>'
        uint64_t q = cur.i;
        for (;;) {
            const char *g = (const char *)memchr(text + q, '>', cur.n - q);
            if (!g) { q = cur.n; break; }
            q = (uint64_t)(g - text);
            if (is_space(text[q - 1])) break;
            q++;
        }
        R.se = q;
        cur.i = q;
        recs.push_back(std::move(R));
    }
    r.n_chr = 0; r.sum_length = 0;
    r.anchor.clear(); r.chr_size.clear(); r.rc_offset.clear(); r.names.clear(); r.blocks.clear(); r.sites.clear();
    r.ccgg_index.assign(P.rrbs ? P.max_seedseg_num : 0, {});
    std::vector<Packed> pk(recs.size());
    {
        // records of 32 MB and more one after another, every thread on chunks of the record; the small ones (contigs, plasmids) by a pool, one thread each
        const unsigned ncpu = n < (uint64_t)BSX_PACK_MINTEXT ? 1u : std::max(1u, std::min(32u, bsx_usable_cpus()));
        std::vector<size_t> small;
        for (size_t i = 0; i < recs.size(); i++) {
            if (recs[i].se - recs[i].sb >= (uint64_t)BSX_PACK_BIG && ncpu > 1) pack_record(P, text, recs[i], (uint32_t)i, code_f, code_r, cls, pk[i], ncpu);
            else small.push_back(i);
        }
        pfor(small.size(), ncpu, [&](size_t k) { pack_record(P, text, recs[small[k]], (uint32_t)small[k], code_f, code_r, cls, pk[small[k]], 1); });
    }
    std::vector<std::vector<uint32_t>> fw, rc;
    for (size_t i = 0; i < recs.size(); i++) {
        Packed &o = pk[i];
        if (o.rc != BSX_OK) return o.rc;
        r.names.push_back(recs[i].name);
        r.chr_size.push_back(o.L);
        r.rc_offset.push_back(o.padded);
        r.blocks.insert(r.blocks.end(), o.blocks.begin(), o.blocks.end());
        fw.push_back(std::move(o.f));
        rc.push_back(std::move(o.c));
        r.n_chr++;
        r.sum_length += o.L;
        if (P.rrbs) {
            for (int sg = 0; sg < P.max_seedseg_num; sg++) { r.ccgg_index[sg].push_back(std::move(o.ccgg_f[sg])); r.ccgg_index[sg].push_back(std::move(o.ccgg_r[sg])); }
            r.sites.push_back(std::move(o.sites));
        }
    }
    std::sort(r.blocks.begin(), r.blocks.end(), [](const Block &a, const Block &b) { return a.id < b.id || (a.id == b.id && a.begin < b.begin); });
    // concatenate with margins (dbseq.cpp:252-273); total nt coordinate space must stay below 2^32
    uint64_t words = 0;
    r.anchor.assign(1, BSX_REF_MARGIN * BSX_SEGLEN);
    for (uint32_t c = 0; c < r.n_chr; c++) {
        words += fw[c].size();
        if ((words + 2 * BSX_REF_MARGIN) * BSX_SEGLEN >= 0xFFFFFFFFull) return BSX_ERR_LIMIT;
        r.anchor.push_back((uint32_t)((words + BSX_REF_MARGIN) * BSX_SEGLEN));
    }
    r.n_words = words + 2 * BSX_REF_MARGIN;
    refcat.assign(r.n_words, 0);   // margins are zero here (uninitialised memory in the reference; never observable)
    crefcat.assign(r.n_words, 0);
    uint64_t o = BSX_REF_MARGIN;
    for (uint32_t c = 0; c < r.n_chr; c++) {
        std::copy(fw[c].begin(), fw[c].end(), refcat.begin() + o);
        std::copy(rc[c].begin(), rc[c].end(), crefcat.begin() + o);
        o += fw[c].size();
    }
    return BSX_OK;
}
