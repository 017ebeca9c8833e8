// White-box harness around the REAL reference objects (TEST INFRASTRUCTURE ONLY).
//
// This file is OUR code.  It is compiled by oracle/Makefile (`make -C oracle ref`) against the
// reference's own object files, built from the sources where they lie under /root/reference, into
// oracle/_ref/libbsmapref.so.  It exposes the reference's internals through a small C ABI so that
// the tests can (1) pin the plain-C restatement in oracle/bsx_oracle.c and (2) generate the golden
// vectors under tests/golden/.  Nothing here is ever linked into the product library.
//
// Reference entry points driven (file:line in /root/reference):
//   Param setters / InitMapping           param.cpp:85-121, main.cpp:234-289 (option order)
//   RefSeq::Run_ConvertBinseq             dbseq.cpp:215
//   RefSeq::CreateIndex                   dbseq.cpp:516
//   SingleAlign::FilterReads / RunAlign   align.cpp:579, align.cpp:435
//   SingleAlign::StringAlign              align.cpp:610
//   PairAlign::RunAlign / Do_Batch body   pairs.cpp:137, pairs.cpp:192-218
#define protected public
#define private public
#include "pairs.h"
#undef protected
#undef private
#include <cstring>
#include <stdint.h>

Param param;  // the reference expects this global (main.cpp:23)

static RefSeq *g_ref = 0;
static SingleAlign *g_sa = 0;
static PairAlign *g_pa = 0;

extern "C" {

struct bsref_params {
    int seed_size;        // -s   (<=0: leave default)
    int index_interval;   // -I   (<=0: leave default)
    int max_snp_num;      // -v
    int max_num_hits;     // -w   (<=0: leave default MAXHITS)
    int chains;           // -n
    int pairend;          // -b given
    int min_insert;       // -m
    int max_insert;       // -x
    int report_repeat_hits; // -r
    int randseed;         // -S
    int qual_threshold;   // -q
    int zero_qual;        // -z  (<=0: leave default '!')
    int max_ns;           // -f  (<0: leave default)
    int out_sam;          // 0 BSP, 1 SAM
    int out_unmap;        // -u
    int out_ref;          // -R
    int max_readlen;      // -L  (<=0: leave default 144)
    const char *digest;   // -D  (NULL: WGBS)
    const char *adapters[10]; // -A
    int n_adapter;
    char read_nt, ref_nt; // -M  (0: leave default TC)
};

// Mirrors the order of effects a command line "-D .. -s .. -I .. (rest)" has in mGetOptions:
// -D forces seed 12 / interval 1 and later -s/-I are overridden while RRBS_flag is set.
int bsref_init(const bsref_params *p)
{
    param = Param();
    if (p->read_nt && p->ref_nt) param.SetAlign(p->read_nt, p->ref_nt);
    if (p->digest) param.SetDigestionSite(p->digest);
    if (p->seed_size > 0) { param.SetSeedSize(p->seed_size); if (param.RRBS_flag) param.SetSeedSize(12); }
    if (p->index_interval > 0) { param.index_interval = p->index_interval; if (param.RRBS_flag) param.index_interval = 1; }
    param.max_snp_num = p->max_snp_num;
    if (p->max_num_hits > 0) param.max_num_hits = p->max_num_hits;
    param.chains = (p->chains != 0);
    param.pairend = p->pairend;
    param.min_insert = p->min_insert;
    param.max_insert = p->max_insert;
    param.report_repeat_hits = p->report_repeat_hits;
    param.randseed = p->randseed;
    param.qual_threshold = p->qual_threshold;
    if (p->zero_qual > 0) param.zero_qual = p->zero_qual;
    if (p->max_ns >= 0) param.max_ns = p->max_ns;
    param.out_sam = p->out_sam;
    param.out_unmap = p->out_unmap;
    param.out_ref = p->out_ref;
    if (p->max_readlen > 0) param.max_readlen = p->max_readlen;
    param.n_adapter = 0;
    for (int i = 0; i < p->n_adapter && i < 10; i++) param.adapter[param.n_adapter++] = p->adapters[i];
    param.InitMapping();
    return 0;
}

int bsref_load(const char *fasta)
{
    ifstream fin(fasta);
    if (!fin) return -1;
    delete g_sa; g_sa = 0;
    delete g_pa; g_pa = 0;
    g_ref = new RefSeq();   // previous one (if any) is leaked on purpose: test process only
    g_ref->n_CCGG = 0;
    g_ref->Run_ConvertBinseq(fin);
    g_ref->CreateIndex();
    return 0;
}

// ---- packed reference ------------------------------------------------------------------------
uint32_t bsref_n_chr() { return g_ref->total_num; }
uint64_t bsref_n_words() { uint64_t s = 0; for (int i = 0; i < g_ref->total_num; i++) s += g_ref->bfa[i * 2].n; return s + 2 * REF_MARGIN; }
const uint32_t *bsref_refcat() { return g_ref->refcat; }
const uint32_t *bsref_crefcat() { return g_ref->crefcat; }
uint32_t bsref_anchor(int i) { return g_ref->ref_anchor[i]; }
uint32_t bsref_chr_size(int c) { return g_ref->title[c * 2].size; }
uint32_t bsref_chr_rc_offset(int c) { return g_ref->title[c * 2].rc_offset; }
const char *bsref_chr_name(int c) { return g_ref->title[c * 2].name.c_str(); }
uint32_t bsref_n_blocks() { return g_ref->_blocks.size(); }
void bsref_block(uint32_t i, uint32_t *id, uint32_t *begin, uint32_t *end)
{ *id = g_ref->_blocks[i].id; *begin = g_ref->_blocks[i].begin; *end = g_ref->_blocks[i].end; }

// ---- seed index --------------------------------------------------------------------------------
uint32_t bsref_total_kmers() { return g_ref->total_kmers; }
// WGBS bucket: returns N (total entries), *n_fwd, *entries -> first entry (dbseq.cpp:381-382,464-465)
uint32_t bsref_bucket(uint32_t key, uint32_t *n_fwd, const uint32_t **entries)
{
    NewIndex u = g_ref->index2[key];
    if (!u) { *n_fwd = 0; *entries = 0; return 0; }
    *n_fwd = u[1] - 2; *entries = u + 2; return u[0] - 2;
}
// RRBS bucket: entries are Hit{chr|seg<<16|dir<<24, loc} (dbseq.cpp:421,429)
uint32_t bsref_rrbs_bucket(uint32_t key, const uint32_t **hits)
{
    KmerLoc *z = g_ref->index + key;
    *hits = z->n1 ? (const uint32_t *)z->loc1 : 0;
    return z->n1;
}
uint32_t bsref_n_sites(int c) { return g_ref->CCGG_sites[c].size(); }
const uint32_t *bsref_sites(int c) { return g_ref->CCGG_sites[c].empty() ? 0 : &g_ref->CCGG_sites[c][0]; }
uint32_t bsref_xt(uint32_t x) { return param.XT(x); }
int bsref_profile_a(int seg, int phase) { return param.profile[seg][phase].a; }
uint32_t bsref_myrand(int index) { uint32_t s = 1; return myrand(index, &s); }

// ---- single-end ------------------------------------------------------------------------------
struct bsref_read_state {
    int filtered;            // FilterReads() result (1 = rejected)
    int len;                 // length after trimming
    int raw_len;
    int read_max_snp_num;
    int seedseg_num;
    int flag_chain, cflag_chain;
    int seed_start_array[16], cseed_start_array[16];
    int seedindex[16], cseedindex[16];          // segment order (.second)
    uint32_t seedcount[16], cseedcount[16];     // .first
    int n_hit[16], n_chit[16];
    uint32_t snp_thres;
};

static void fill_state(SingleAlign &a, int filtered, bsref_read_state *st)
{
    memset(st, 0, sizeof(*st));
    st->filtered = filtered;
    st->len = a._pread->seq.size();
    st->raw_len = a.raw_readlen;
    if (filtered) return;
    st->read_max_snp_num = a.read_max_snp_num;
    st->seedseg_num = a.seedseg_num;
    st->flag_chain = a.flag_chain; st->cflag_chain = a.cflag_chain;
    for (int i = 0; i < 16; i++) { st->seed_start_array[i] = a.seed_start_array[i]; st->cseed_start_array[i] = a.cseed_start_array[i]; }
    if (a.flag_chain) for (size_t i = 0; i < a.seedindex.size() && i < 16; i++) { st->seedindex[i] = a.seedindex[i].second; st->seedcount[i] = a.seedindex[i].first; }
    if (a.cflag_chain) for (size_t i = 0; i < a.cseedindex.size() && i < 16; i++) { st->cseedindex[i] = a.cseedindex[i].second; st->cseedcount[i] = a.cseedindex[i].first; }
    for (int i = 0; i <= param.max_snp_num && i < 16; i++) { st->n_hit[i] = a._cur_n_hit[i]; st->n_chit[i] = a._cur_n_chit[i]; }
    st->snp_thres = a.snp_thres;
}

// The reference never initialises these members (align.h:82-91); give the first read a defined start.
static void zero_leak_state(SingleAlign &a)
{
    a.seed_start_offset = a.cseed_start_offset = 0;
    memset(a.seed_array, 0, sizeof(a.seed_array)); memset(a.cseed_array, 0, sizeof(a.cseed_array));
    memset(a.seed_start_array, 0, sizeof(a.seed_start_array)); memset(a.cseed_start_array, 0, sizeof(a.cseed_start_array));
}

static void set_read(SingleAlign &a, uint32_t index, int readset, const char *name, const char *seq, const char *qual)
{
    a.mreads.resize(1);
    ReadInf &r = a.mreads[0];
    r.index = index; r.readset = readset; r.name = name; r.seq = seq; r.qual = qual;
    // reads.cpp:115-117: reads are truncated at load time
    if ((int)r.seq.size() > param.max_readlen) { r.seq.erase(param.max_readlen); r.qual.erase(param.max_readlen); }
    a.num_reads = 1;
    a._pread = a.mreads.begin();
}

// one read through the body of SingleAlign::Do_Batch (align.cpp:591-606); the SingleAlign object
// persists between calls so cross-read state leaks of the reference are reproduced in call order.
// out_line (may be NULL) receives the formatted SAM/BSP text (may be empty).
int bsref_se_align(uint32_t index, const char *name, const char *seq, const char *qual,
                   bsref_read_state *st, char *out_line, int out_cap)
{
    if (!g_sa) { g_sa = new SingleAlign(); zero_leak_state(*g_sa); }
    SingleAlign &a = *g_sa;
    set_read(a, index, 0, name, seq, qual);
    a._str_align.clear();
    int f = a.FilterReads();
    if (f) {
        if (param.report_repeat_hits) a.s_OutHit(0, -1, 0, a.hits[0], 0, *g_ref, a._str_align);
        // NB the reference emits nothing for filtered reads under -r 0 (align.cpp:599)
    } else {
        a.RunAlign(*g_ref);
    }
    fill_state(a, f, st);
    if (!f) a.StringAlign(*g_ref, a._str_align);
    if (out_line) { strncpy(out_line, a._str_align.c_str(), out_cap - 1); out_line[out_cap - 1] = 0; }
    return f;
}
// hit lists of the last SE read: orient 0 -> hits, 1 -> chits; returns pointer to Hit{chr,loc} pairs
const uint32_t *bsref_se_hits(int orient, int w) { return (const uint32_t *)(orient ? g_sa->chits[w] : g_sa->hits[w]); }
uint32_t bsref_se_n_aligned() { return g_sa ? g_sa->n_aligned : 0; }

// ---- paired-end ------------------------------------------------------------------------------
struct bsref_pair_state {
    int paired;                 // PairAlign::RunAlign return (0 = none, else level+1); 0 if a mate was filtered
    int tmp;                    // StringAlignPair return when paired (0 printed pair, 1 fall through to unpaired)
    uint32_t n_pairs[31];       // _cur_n_hits[na+nb]
    bsref_read_state a, b;
};

int bsref_pe_align(uint32_t index, const char *name_a, const char *seq_a, const char *qual_a,
                   const char *name_b, const char *seq_b, const char *qual_b,
                   bsref_pair_state *st, char *out_line, int out_cap, char *out_unpair, int unpair_cap)
{
    if (!g_pa) { g_pa = new PairAlign(); zero_leak_state(g_pa->_sa); zero_leak_state(g_pa->_sb); }
    PairAlign &p = *g_pa;
    set_read(p._sa, index, 1, name_a, seq_a, qual_a);
    set_read(p._sb, index, 2, name_b, seq_b, qual_b);
    p.num_reads = 1;
    p._str_align.clear(); p._str_align_unpair.clear();
    memset(st, 0, sizeof(*st));
    // body of PairAlign::Do_Batch (pairs.cpp:203-217)
    int filter1 = p._sa.FilterReads(), filter2 = p._sb.FilterReads();
    p.FixPairReadName();
    int paired, tmp = 0;
    for (int i = 0; i < 31; i++) p._cur_n_hits[i] = 0;
    if (filter1 == 0 && filter2 == 0) paired = p.RunAlign(*g_ref);
    else { paired = 0; if (filter1 == 0) p._sa.RunAlign(*g_ref); if (filter2 == 0) p._sb.RunAlign(*g_ref); }
    fill_state(p._sa, filter1, &st->a);
    fill_state(p._sb, filter2, &st->b);
    st->paired = paired;
    if (filter1 == 0 && filter2 == 0) for (int i = 0; i <= param.max_snp_num * 2; i++) st->n_pairs[i] = p._cur_n_hits[i];
    if (paired) tmp = p.StringAlignPair(*g_ref, p._str_align);
    st->tmp = tmp;
    if (tmp == 1 || paired == 0) {
        if (param.out_sam) p.StringAlignUnpair(filter1, filter2, *g_ref, p._str_align);
        else p.StringAlignUnpair(filter1, filter2, *g_ref, p._str_align_unpair);
    }
    if (out_line) { strncpy(out_line, p._str_align.c_str(), out_cap - 1); out_line[out_cap - 1] = 0; }
    if (out_unpair) { strncpy(out_unpair, p._str_align_unpair.c_str(), unpair_cap - 1); out_unpair[unpair_cap - 1] = 0; }
    return paired;
}
const uint32_t *bsref_pe_hits(int mate, int orient, int w)
{
    SingleAlign &s = mate ? g_pa->_sb : g_pa->_sa;
    return (const uint32_t *)(orient ? s.chits[w] : s.hits[w]);
}
// PairHit layout: {u16 chain; u8 na; u8 nb; int insert; Hit a; Hit b} = 24 bytes (pairs.h:13-20)
const void *bsref_pe_pairs(int w) { return g_pa->pairhits[w]; }
int bsref_sizeof_pairhit() { return sizeof(PairHit); }
void bsref_pe_counters(uint32_t *pairs, uint32_t *a, uint32_t *b)
{ *pairs = g_pa ? g_pa->n_aligned_pairs : 0; *a = g_pa ? g_pa->n_aligned_a : 0; *b = g_pa ? g_pa->n_aligned_b : 0; }

} // extern "C"
