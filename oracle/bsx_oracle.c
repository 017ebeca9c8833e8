/* bsx_oracle.c — CPU restatement of the BSMAP v2.6 hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain C99 restatement, function by function, of the reference algorithm on the path
 *   reference packer -> seed index -> read packer -> seed planner -> candidate scan + mismatch
 *   extension -> per-read driver -> PE pairing -> trimming -> RRBS variants.
 * It is the parity checker for the HIP product and the timed "port" CPU baseline in bench.py; the
 * product never links or calls it.  Parity status: PINNED — tests/test_oracle_vs_reference.py compares
 * every stage with the real reference (oracle/_ref/libbsmapref.so, built from /root/reference) and
 * tests/test_oracle_golden.py compares it with the committed vectors in tests/golden/.
 *
 * Citations are file:line in /root/reference.
 */
#define _GNU_SOURCE
#include "bsx_oracle.h"
#include <ctype.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define SEGLEN 16
#define FIXSIZE (SEGLEN * BSO_FIXELEMENT)

/* ------------------------------------------------------------------------------------------------
 * a1/a2/a4: parameters, alphabet, 3-letter hash, seed profile
 * ---------------------------------------------------------------------------------------------- */
void bso_params_default(bso_params *p) /* param.cpp:6-83 */
{
    memset(p, 0, sizeof(*p));
    p->seed_size = 16;
    p->index_interval = 4;
    p->max_snp_num = 2;
    p->max_num_hits = BSO_MAXHITS;
    p->min_insert = 28;
    p->max_insert = 500;
    p->report_repeat_hits = 1;
    p->zero_qual = '!';
    p->max_ns = 5;
    p->max_readlen = (BSO_FIXELEMENT - 1) * 16;
    p->read_nt = 'T';
    p->ref_nt = 'C';
}

int bso_params_set_digest(bso_params *p, const char *a) /* param.cpp:95-106 */
{
    const char *d = strchr(a, '-');
    if (!d) return -1;
    p->digest_pos = (int)(d - a);
    size_t n = strlen(a);
    if (n - 1 >= sizeof(p->digest_site)) return -1;
    memcpy(p->digest_site, a, p->digest_pos);
    strcpy(p->digest_site + p->digest_pos, d + 1);
    p->rrbs = 1;
    p->index_interval = 1;
    p->seed_size = 12;
    return 0;
}

static const char nt_code[4] = {'A', 'C', 'G', 'T'};
static int alphabet0(int c) { c = toupper(c); return c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 0; }
static int is_acgt(int c) { c = toupper(c); return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

int bso_params_finish(bso_params *p)
{
    /* Param::SetAlign, param.cpp:187-231 */
    int i, j;
    uint8_t bit_nt[4], tmp = 0;
    int read_nt = toupper(p->read_nt), ref_nt = toupper(p->ref_nt);
    if (!is_acgt(read_nt) || !is_acgt(ref_nt) || read_nt == ref_nt) return -1;
    for (i = 0; i < 4; i++) bit_nt[i] = 100;
    bit_nt[alphabet0(read_nt)] = 3;
    bit_nt[alphabet0(ref_nt)] = 1;
    for (i = 0; i < 4; i++)
        if (nt_code[i] != ref_nt && nt_code[i] != read_nt) { bit_nt[i] = tmp; tmp = 2; }
    for (i = 0; i < 256; i++) p->alphabet[i] = bit_nt[0];
    p->alphabet['c'] = p->alphabet['C'] = bit_nt[1];
    p->alphabet['g'] = p->alphabet['G'] = bit_nt[2];
    p->alphabet['t'] = p->alphabet['T'] = bit_nt[3];
    for (i = 0; i < 256; i++) p->rev_alphabet[i] = bit_nt[3];
    p->rev_alphabet['c'] = p->rev_alphabet['C'] = bit_nt[2];
    p->rev_alphabet['g'] = p->rev_alphabet['G'] = bit_nt[1];
    p->rev_alphabet['t'] = p->rev_alphabet['T'] = bit_nt[0];
    memset(p->reg_alphabet, 0, 256); /* param.cpp:153-163 */
    for (i = 0; i < 8; i++) p->reg_alphabet[(unsigned char)"ACGTacgt"[i]] = 3;
    strcpy(p->useful_nt, "ACGTacgt");
    for (i = 0; i < 4; i++) { p->useful_nt[bit_nt[i]] = nt_code[i]; p->useful_nt[bit_nt[i] + 4] = (char)tolower(nt_code[i]); }
    /* Param::SetSeedSize, param.cpp:108-120 */
    p->seed_bits = 0;
    for (i = 0; i < p->seed_size; i++) p->seed_bits |= 0x3u << (i * 2);
    /* Param::InitMapping, param.cpp:85-93 */
    for (i = 0; i < p->index_interval; i++)
        for (j = 0; j <= BSO_MAXSNPS; j++)
            p->profile_a[j][i] = (uint8_t)(((j * p->seed_size + i + p->index_interval - 1) / p->index_interval) * p->index_interval);
    p->max_seedseg_num = (BSO_FIXELEMENT - 1) * 16 / p->seed_size; /* dbseq.cpp:217 */
    p->total_kmers = 1;
    for (i = 0; i < p->seed_size; i++) p->total_kmers *= 3; /* dbseq.cpp:314 */
    return 0;
}

/* Param::BuildMismatchTable + XT, param.cpp:122-137, param.h:123 — computed arithmetically:
 * per nt collapse code 3 -> 1, then read the 16 2-bit digits as a base-3 number, first nt most significant */
static inline uint32_t xt16(uint32_t i)
{
    uint32_t TT = ((~((i << 1) & i)) | 0x5555u) & i & 0xFFFFu;
    uint32_t n = 0;
    for (int j = 7; j >= 0; j--) n = n * 3 + ((TT >> (j * 2)) & 3);
    return n;
}
static uint16_t g_T[65536];
static pthread_once_t g_T_once = PTHREAD_ONCE_INIT;
static void build_T(void) { for (uint32_t i = 0; i < 65536; i++) g_T[i] = (uint16_t)xt16(i); }
static inline uint32_t XT(uint32_t tt) { return (uint32_t)g_T[tt & 0xFFFF] + (uint32_t)g_T[tt >> 16] * 6561u; }
uint32_t bso_xt(const bso_params *p, uint32_t s) { (void)p; pthread_once(&g_T_once, build_T); return XT(s); }

/* Param::XC64 / XM64, param.h:126,139-147 */
static inline uint64_t XC64(uint64_t tt) { return ((~tt) << 1) | tt | 0x5555555555555555ULL; }
static inline uint32_t XM64(uint64_t tt) { return (uint32_t)__builtin_popcountll((tt | (tt >> 1)) & 0x5555555555555555ULL); }

/* ------------------------------------------------------------------------------------------------
 * a5: reference packer
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint32_t id, begin, end; } blk_t;
static int blk_cmp(const void *x, const void *y)
{
    const blk_t *a = x, *b = y; /* BlockComp, dbseq.cpp:213 */
    if (a->id != b->id) return a->id < b->id ? -1 : 1;
    if (a->begin != b->begin) return a->begin < b->begin ? -1 : 1;
    return 0;
}
static int is_ws(int c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }
static int in_set(const char *set, int c) { return c && strchr(set, c) != NULL; }

typedef struct { uint32_t *v; uint32_t n, cap; } u32vec;
static void u32_push(u32vec *v, uint32_t x)
{
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 64; v->v = realloc(v->v, (size_t)v->cap * 4); }
    v->v[v->n++] = x;
}

/* per-chromosome RRBS scratch kept until the index is built (RefSeq::CCGG_index, dbseq.h:106) */
typedef struct { u32vec bsw[50], bsc[50]; } ccgg_chr_t;
struct bso_ref_priv { ccgg_chr_t *ccgg; int borrowed_ref, borrowed_index; };
#define PRIV(r) ((struct bso_ref_priv *)((r)->names[(r)->n_chr]))

static void find_ccgg(const bso_params *p, bso_ref *r, ccgg_chr_t *cc, const char *seq_upper, uint64_t seqcap,
                      uint32_t length, uint32_t c) /* RefSeq::find_CCGG, dbseq.cpp:144-211 */
{
    u32vec sites = {0};
    size_t dl = strlen(p->digest_site);
    uint32_t tmp_offset = r->rc_offset[c] - p->seed_size, tmp_max = r->chr_size[c] - p->seed_size;
    /* _seq.find over the whole buffer; stops at the first match position >= _length */
    for (uint64_t right = 0; right + dl <= seqcap && right < length; right++)
        if (memcmp(seq_upper + right, p->digest_site, dl) == 0) u32_push(&sites, (uint32_t)right + p->digest_pos);
    r->sites[c] = sites.v; r->n_sites[c] = sites.n;
    if (sites.n > 1) {
        for (uint32_t k = 0; k + 1 < sites.n; k++)
            if (sites.v[k + 1] - sites.v[k] <= (uint32_t)p->max_insert) {
                int i, seedloc;
                for (i = 0, seedloc = (int)sites.v[k]; i < p->max_seedseg_num && (uint32_t)seedloc <= tmp_max; i++, seedloc += p->seed_size)
                    u32_push(&cc->bsw[i], (uint32_t)seedloc);
            }
        for (uint32_t k = 1; k < sites.n; k++)
            if (sites.v[k] - sites.v[k - 1] <= (uint32_t)p->max_insert) {
                int i, seedloc;
                for (i = 0, seedloc = (int)((size_t)sites.v[k] + dl - 2 * p->digest_pos - p->seed_size); i < p->max_seedseg_num && seedloc >= 0; i++, seedloc -= p->seed_size)
                    u32_push(&cc->bsc[i], tmp_offset - (uint32_t)seedloc);
            }
    }
}

bso_ref *bso_ref_from_fasta_text(const bso_params *p, const char *text, uint64_t n)
{
    /* RefSeq::Run_ConvertBinseq, dbseq.cpp:215-282.  The reference parses with ifstream>> tokens
     * (LoadNextSeq, dbseq.cpp:18-54): first non-blank char is consumed unchecked, next token is the
     * name, rest of that line is dropped, then whitespace-separated tokens are concatenated until a
     * token starts with '>'. */
    pthread_once(&g_T_once, build_T);
    bso_ref *r = calloc(1, sizeof(*r));
    uint32_t cap_chr = 64;
    r->chr_size = malloc(cap_chr * 4); r->rc_offset = malloc(cap_chr * 4); r->names = malloc((cap_chr + 1) * sizeof(char *));
    r->sites = calloc(cap_chr, sizeof(uint32_t *)); r->n_sites = calloc(cap_chr, 4);
    uint32_t **fw = malloc(cap_chr * sizeof(*fw)), **rc = malloc(cap_chr * sizeof(*rc)), *nw = malloc(cap_chr * 4);
    ccgg_chr_t *ccgg = NULL;
    blk_t *blocks = NULL; uint32_t nb = 0, capb = 0;
    char *seq = NULL; uint64_t seqcap = 0; /* the reference's _seq buffer: never cleared between records */
    uint64_t pos = 0;
    const char *useful = p->useful_nt, *nx = "NXnx";
    while (1) {
        while (pos < n && is_ws(text[pos])) pos++;
        if (pos >= n) break;
        pos++; /* fin>>c */
        while (pos < n && is_ws(text[pos])) pos++;
        uint64_t ns = pos;
        while (pos < n && !is_ws(text[pos])) pos++;
        char *name = strndup(text + ns, pos - ns);
        while (pos < n && text[pos] != '\n') pos++; /* getline */
        if (pos < n) pos++;
        uint64_t length = 0;
        while (1) {
            while (pos < n && is_ws(text[pos])) pos++;
            if (pos >= n || text[pos] == '>') break;
            uint64_t ts = pos;
            while (pos < n && !is_ws(text[pos])) pos++;
            uint64_t tl = pos - ts;
            if (length + tl + 64 > seqcap) { uint64_t nc = (length + tl + 64) * 2; seq = realloc(seq, nc); memset(seq + seqcap, 0, nc - seqcap); seqcap = nc; }
            memcpy(seq + length, text + ts, tl);
            length += tl;
        }
        if (length == 0) { free(name); break; } /* while(LoadNextSeq(fin)) stops at an empty record */
        uint32_t c = r->n_chr;
        if (c + 1 >= cap_chr) {
            cap_chr *= 2;
            r->chr_size = realloc(r->chr_size, cap_chr * 4); r->rc_offset = realloc(r->rc_offset, cap_chr * 4);
            r->names = realloc(r->names, (cap_chr + 1) * sizeof(char *));
            r->sites = realloc(r->sites, cap_chr * sizeof(uint32_t *)); r->n_sites = realloc(r->n_sites, cap_chr * 4);
            fw = realloc(fw, cap_chr * sizeof(*fw)); rc = realloc(rc, cap_chr * sizeof(*rc)); nw = realloc(nw, cap_chr * 4);
        }
        r->sites[c] = NULL; r->n_sites[c] = 0;
        r->names[c] = name;
        r->chr_size[c] = (uint32_t)length;
        uint32_t an = ((uint32_t)length + (SEGLEN - 1)) / SEGLEN + 2; /* BinSeq, dbseq.cpp:58-83 */
        r->rc_offset[c] = an * SEGLEN;
        uint64_t tot = (uint64_t)an * SEGLEN;
        if (tot + 64 > seqcap) { uint64_t nc = (tot + 64) * 2; seq = realloc(seq, nc); memset(seq + seqcap, 0, nc - seqcap); seqcap = nc; }
        memset(seq + length, 'N', tot - length);
        uint32_t *s = malloc((size_t)an * 4), *cs = malloc((size_t)an * 4);
        for (uint32_t i = 0; i < an; i++) {
            uint32_t w = 0;
            for (int j = 0; j < SEGLEN; j++) w = (w << 2) | p->alphabet[(unsigned char)seq[(uint64_t)i * SEGLEN + j]];
            s[i] = w;
        }
        /* UnmaskRegion, dbseq.cpp:114-142 (ids are 2c and 2c+1) */
        {
            blk_t b, cb; b.id = 2 * c; cb.id = 2 * c + 1;
            uint32_t total_len = an * SEGLEN;
            b.begin = b.end = 0;
            while (b.end < length) {
                uint64_t q = b.end;
                while (q < seqcap && !in_set(useful, seq[q])) q++; /* find_first_of(useful_nt, b.end) over the whole buffer */
                if (q >= seqcap || q > length) break;
                b.begin = (uint32_t)q;
                while (q < seqcap && !in_set(nx, seq[q])) q++;
                b.end = (q <= length) ? (uint32_t)q : (uint32_t)length;
                if (b.end - b.begin < 30) continue;
                if (nb && b.id == blocks[nb - 1].id && b.begin - blocks[nb - 1].end < 5) blocks[nb - 1].end = b.end; /* dead in practice: last pushed is the rc twin */
                else {
                    if (nb + 2 > capb) { capb = capb ? capb * 2 : 256; blocks = realloc(blocks, capb * sizeof(blk_t)); }
                    blocks[nb++] = b;
                    cb.begin = total_len - b.end; cb.end = total_len - b.begin;
                    blocks[nb++] = cb;
                }
            }
        }
        /* cBinSeq, dbseq.cpp:85-111 */
        for (uint32_t i = 0; i < an; i++) {
            uint32_t w = 0; uint64_t q = (uint64_t)an * SEGLEN - 1 - (uint64_t)i * SEGLEN;
            for (int j = 0; j < SEGLEN; j++) w = (w << 2) | p->rev_alphabet[(unsigned char)seq[q - j]];
            cs[i] = w;
        }
        fw[c] = s; rc[c] = cs; nw[c] = an;
        r->n_chr++;
        r->sum_length += length;
        if (p->rrbs) {
            ccgg = realloc(ccgg, r->n_chr * sizeof(ccgg_chr_t));
            memset(&ccgg[c], 0, sizeof(ccgg_chr_t));
            for (uint64_t q = 0; q < seqcap; q++) seq[q] = (char)toupper((unsigned char)seq[q]); /* dbseq.cpp:151 */
            find_ccgg(p, r, &ccgg[c], seq, seqcap, (uint32_t)length, c);
        }
    }
    free(seq);
    qsort(blocks, nb, sizeof(blk_t), blk_cmp);
    r->n_blocks = nb;
    r->blk_id = malloc((size_t)(nb ? nb : 1) * 4); r->blk_begin = malloc((size_t)(nb ? nb : 1) * 4); r->blk_end = malloc((size_t)(nb ? nb : 1) * 4);
    for (uint32_t i = 0; i < nb; i++) { r->blk_id[i] = blocks[i].id; r->blk_begin[i] = blocks[i].begin; r->blk_end[i] = blocks[i].end; }
    free(blocks);
    /* dbseq.cpp:252-273 */
    r->anchor = malloc((size_t)(r->n_chr + 1) * 4);
    uint32_t s = 0;
    r->anchor[0] = BSO_REF_MARGIN * SEGLEN;
    for (uint32_t i = 0; i < r->n_chr; i++) { s += nw[i]; r->anchor[i + 1] = (s + BSO_REF_MARGIN) * SEGLEN; }
    r->n_words = (uint64_t)s + 2 * BSO_REF_MARGIN;
    r->refcat = calloc(r->n_words + 16, 4); r->crefcat = calloc(r->n_words + 16, 4); /* margins zero-filled (reference: uninitialised) */
    uint64_t o = BSO_REF_MARGIN;
    for (uint32_t i = 0; i < r->n_chr; i++) {
        memcpy(r->refcat + o, fw[i], (size_t)nw[i] * 4); memcpy(r->crefcat + o, rc[i], (size_t)nw[i] * 4);
        o += nw[i]; free(fw[i]); free(rc[i]);
    }
    free(fw); free(rc); free(nw);
    struct bso_ref_priv *pv = calloc(1, sizeof(*pv));
    pv->ccgg = ccgg;
    r->names[r->n_chr] = (char *)pv;
    return r;
}

bso_ref *bso_ref_from_fasta_file(const bso_params *p, const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    char *buf = malloc((size_t)n + 1);
    if (fread(buf, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(buf); return NULL; }
    fclose(f);
    bso_ref *r = bso_ref_from_fasta_text(p, buf, (uint64_t)n);
    free(buf);
    return r;
}

bso_ref *bso_ref_wrap(uint32_t n_chr, uint64_t n_words, uint32_t *refcat, uint32_t *crefcat, uint32_t *anchor,
                      uint32_t *chr_size, uint32_t *rc_offset)
{
    bso_ref *r = calloc(1, sizeof(*r));
    r->n_chr = n_chr; r->n_words = n_words; r->refcat = refcat; r->crefcat = crefcat;
    r->anchor = anchor; r->chr_size = chr_size; r->rc_offset = rc_offset;
    r->names = calloc(n_chr + 1, sizeof(char *));
    struct bso_ref_priv *pv = calloc(1, sizeof(*pv));
    pv->borrowed_ref = 1;
    r->names[n_chr] = (char *)pv;
    return r;
}

void bso_index_attach(bso_ref *r, uint32_t total_kmers, uint32_t *bucket_off, uint32_t *bucket_nfwd, uint32_t *entries, uint64_t n_entries)
{
    r->total_kmers = total_kmers; r->bucket_off = bucket_off; r->bucket_nfwd = bucket_nfwd; r->entries = entries; r->n_entries = n_entries;
    PRIV(r)->borrowed_index = 1;
}

void bso_ref_free(bso_ref *r)
{
    if (!r) return;
    struct bso_ref_priv *pv = PRIV(r);
    if (!pv->borrowed_ref) {
        free(r->refcat); free(r->crefcat); free(r->anchor); free(r->chr_size); free(r->rc_offset);
        for (uint32_t i = 0; i < r->n_chr; i++) { free(r->names[i]); if (r->sites) free(r->sites[i]); }
    }
    if (!pv->borrowed_index) { free(r->bucket_off); free(r->bucket_nfwd); free(r->entries); free(r->rrbs_entries); }
    free(r->blk_id); free(r->blk_begin); free(r->blk_end); free(r->sites); free(r->n_sites);
    if (pv->ccgg) {
        for (uint32_t c = 0; c < r->n_chr; c++) for (int i = 0; i < 50; i++) { free(pv->ccgg[c].bsw[i].v); free(pv->ccgg[c].bsc[i].v); }
        free(pv->ccgg);
    }
    free(pv); free(r->names); free(r);
}

/* ------------------------------------------------------------------------------------------------
 * a6: seed index
 * ---------------------------------------------------------------------------------------------- */
/* RefSeq::s_MakeSeed_1, dbseq.cpp:286-291: seed of seed_size nt starting at nt `loc` of strand array m */
static inline uint32_t make_seed(const bso_params *p, const uint32_t *m, uint32_t loc)
{
    const uint32_t *w = m + loc / SEGLEN;
    uint32_t a = 64 - p->seed_size * 2 - (loc % SEGLEN) * 2;
    return XT((uint32_t)(((((uint64_t)w[0] << 32) | w[1]) >> a) & p->seed_bits));
}

int bso_index_build(const bso_params *p, bso_ref *r) /* RefSeq::CreateIndex, dbseq.cpp:516-539 */
{
    pthread_once(&g_T_once, build_T);
    uint32_t K = p->total_kmers, I = p->index_interval, S = p->seed_size;
    r->total_kmers = K;
    uint32_t *cnt = calloc((size_t)K + 1, 4), *nf = calloc((size_t)K, 4);
    if (p->rrbs) {
        /* t_CalKmerFreq_ab RRBS branch, dbseq.cpp:332-347; t_CreateIndex_ab, dbseq.cpp:418-438 */
        ccgg_chr_t *cc = PRIV(r)->ccgg;
        for (int pass = 0; pass < 2; pass++) {
            for (int j = 0; j < p->max_seedseg_num; j++)
                for (uint32_t chr = 0; chr < 2 * r->n_chr; chr++) {
                    const uint32_t *m = ((chr & 1) ? r->crefcat : r->refcat) + r->anchor[chr / 2] / SEGLEN;
                    u32vec *own = (chr & 1) ? &cc[chr / 2].bsc[j] : &cc[chr / 2].bsw[j];
                    u32vec *other = (chr & 1) ? &cc[chr / 2].bsw[j] : &cc[chr / 2].bsc[j]; /* CCGG_index[j][chr^1] */
                    for (uint32_t k = 0; k < own->n; k++) {
                        uint32_t key = make_seed(p, m, own->v[k]);
                        if (!pass) cnt[key]++;
                        else { uint32_t q = cnt[key]++; r->rrbs_entries[2 * (size_t)q] = chr | ((uint32_t)j << 16); r->rrbs_entries[2 * (size_t)q + 1] = own->v[k]; }
                    }
                    if (p->pairend || p->chains) {
                        uint32_t tmp_offset = r->rc_offset[chr / 2] - S;
                        for (uint32_t k = 0; k < other->n; k++) {
                            uint32_t loc = tmp_offset - other->v[k], key = make_seed(p, m, loc);
                            if (!pass) cnt[key]++;
                            else { uint32_t q = cnt[key]++; r->rrbs_entries[2 * (size_t)q] = chr | ((uint32_t)j << 16) | 0x1000000u; r->rrbs_entries[2 * (size_t)q + 1] = loc; }
                        }
                    }
                }
            if (!pass) {
                uint32_t acc = 0;
                r->bucket_off = malloc(((size_t)K + 1) * 4);
                for (uint32_t k = 0; k < K; k++) { r->bucket_off[k] = acc; acc += cnt[k]; cnt[k] = r->bucket_off[k]; }
                r->bucket_off[K] = acc; r->n_entries = acc;
                r->rrbs_entries = malloc(((size_t)acc + 1) * 8);
            }
        }
        r->bucket_nfwd = nf;
        free(cnt);
        return 0;
    }
    /* t_CalKmerFreq_ab, dbseq.cpp:349-361 */
    for (uint32_t b = 0; b < r->n_blocks; b++) {
        uint32_t id = r->blk_id[b];
        const uint32_t *m = ((id & 1) ? r->crefcat : r->refcat) + r->anchor[id / 2] / SEGLEN;
        uint32_t i2 = ((r->blk_end[b] - S) / I) * I;
        for (uint32_t i = (r->blk_begin[b] / I) * I; i <= i2; i += I) {
            uint32_t key = make_seed(p, m, i);
            cnt[key]++;
            if (!(id & 1)) nf[key]++;
        }
    }
    /* AllocIndex, dbseq.cpp:365-388 -> CSR offsets */
    r->bucket_off = malloc(((size_t)K + 1) * 4);
    uint64_t acc = 0;
    for (uint32_t k = 0; k < K; k++) { r->bucket_off[k] = (uint32_t)acc; acc += cnt[k]; }
    r->bucket_off[K] = (uint32_t)acc; r->n_entries = acc;
    r->entries = malloc((acc + 16) * 4);
    memset(r->entries + acc, 0, 16 * 4);
    /* t_CreateIndex_ab, dbseq.cpp:439-480: all forward-strand blocks first, then all rc-strand blocks */
    uint32_t *cur = cnt;
    for (uint32_t k = 0; k < K; k++) cur[k] = r->bucket_off[k];
    for (int parity = 0; parity < 2; parity++)
        for (uint32_t b = 0; b < r->n_blocks; b++) {
            uint32_t id = r->blk_id[b];
            if ((int)(id & 1) != parity) continue;
            const uint32_t *m = ((id & 1) ? r->crefcat : r->refcat) + r->anchor[id / 2] / SEGLEN;
            uint32_t i2 = ((r->blk_end[b] - S) / I) * I;
            for (uint32_t loc = (r->blk_begin[b] / I) * I; loc <= i2; loc += I)
                r->entries[cur[make_seed(p, m, loc)]++] = r->anchor[id / 2] + loc; /* hit2int, dbseq.cpp:570 */
        }
    r->bucket_nfwd = nf;
    free(cnt);
    return 0;
}

/* RefSeq::CCGG_seglen, dbseq.cpp:541-567 */
static void ccgg_seglen(const bso_params *p, const bso_ref *r, uint32_t chr, uint32_t pos, int readlen, uint32_t *first, int *second)
{
    uint32_t chr2 = chr / 2;
    const uint32_t *sites = r->sites[chr2];
    int size = (int)r->n_sites[chr2], left = 0, right = size - 1, mid;
    uint32_t midval, seg_start, seg_end = 0;
    size_t dl = strlen(p->digest_site);
    while (left < right - 1) {
        mid = (left + right) / 2;
        if ((midval = sites[mid]) == pos) { left = mid; right = mid + 1; break; }
        else if (midval < pos) left = mid;
        else right = mid;
    }
    seg_start = size > 0 ? sites[left] : 0;
    while (1) {
        /* the reference reads sites[right] before testing right<size (one-past-the-end read = UB); we read 0 there */
        uint32_t sv = (right >= 0 && right < size) ? sites[right] : 0;
        seg_end = (uint32_t)((size_t)sv + dl - (size_t)(p->digest_pos * 2));
        if (seg_end < pos + (uint32_t)readlen && right < size) right++;
        else break;
    }
    *first = seg_start + 1;
    *second = (int)(seg_end - seg_start);
}

/* ------------------------------------------------------------------------------------------------
 * aligner state (SingleAlign, align.h:24-134)
 * ---------------------------------------------------------------------------------------------- */
#define HS_CAP 65536u
typedef bso_hit hit_array[BSO_MAXHITS + 1];
typedef bso_pair pair_array[BSO_MAXHITS + 1];

struct bso_aligner {
    const bso_params *P;
    const bso_ref *R;
    int leak_mode;
    uint32_t bseq[16][10], reg[16][10], cbseq[16][10], creg[16][10];
    uint32_t seeds[BSO_MAXSNPS + 1][16], cseeds[BSO_MAXSNPS + 1][16];
    uint32_t seed_array[144 + 32], cseed_array[144 + 32];
    int n_hit[BSO_MAXSNPS + 1], n_chit[BSO_MAXSNPS + 1];
    uint32_t snp_thres;
    int seed_start_offset, cseed_start_offset;
    int seed_start_array[BSO_MAXSNPS + 1], cseed_start_array[BSO_MAXSNPS + 1];
    uint32_t cseed_offset;
    struct { uint32_t cnt; int idx; } seedindex[16], cseedindex[16];
    hit_array *hits, *chits;
    /* hitset: open-addressing set of (chr>>1, loc), cleared per read (align.cpp:428-433) */
    uint64_t *hs; uint32_t *hs_used; uint32_t hs_n;
    int flag_chain, cflag_chain;
    int raw_readlen, read_max_snp_num, seedseg_num;
    char seq[FIXSIZE + 64], qual[FIXSIZE + 64];
    int len, qlen;
    uint32_t index; int readset;
    uint32_t rand_rseed;
    /* pairing (PairAlign, pairs.h:49-65) — owned by the mate-a aligner */
    pair_array *pairhits; uint32_t n_pairs[2 * BSO_MAXSNPS + 1];
    uint64_t n_lookup, n_cand, sum_w, n_orient;
};

bso_aligner *bso_aligner_new(const bso_params *p, const bso_ref *r, int leak_mode)
{
    pthread_once(&g_T_once, build_T);
    bso_aligner *a = calloc(1, sizeof(*a));
    a->P = p; a->R = r; a->leak_mode = leak_mode;
    a->hits = calloc(BSO_MAXSNPS + 1, sizeof(hit_array));
    a->chits = calloc(BSO_MAXSNPS + 1, sizeof(hit_array));
    a->pairhits = calloc(2 * BSO_MAXSNPS + 1, sizeof(pair_array));
    a->hs = calloc(HS_CAP, 8); a->hs_used = malloc(HS_CAP * 4);
    a->rand_rseed = 12345;
    return a;
}
void bso_aligner_free(bso_aligner *a)
{
    if (!a) return;
    free(a->hits); free(a->chits); free(a->pairhits); free(a->hs); free(a->hs_used); free(a);
}
const bso_hit *bso_se_hits(const bso_aligner *a, int orient, int w) { return orient ? a->chits[w] : a->hits[w]; }
const bso_pair *bso_pe_pairs(const bso_aligner *a, int w) { return a->pairhits[w]; }
void bso_counters(const bso_aligner *a, uint64_t *nl, uint64_t *nc, uint64_t *sw, uint64_t *no)
{ *nl = a->n_lookup; *nc = a->n_cand; *sw = a->sum_w; *no = a->n_orient; }

static inline int hs_insert(bso_aligner *a, uint32_t chrpair, uint32_t loc) /* returns 1 if newly inserted */
{
    uint64_t key = (((uint64_t)chrpair << 32) | loc) + 1;
    uint32_t h = (uint32_t)((key * 0x9E3779B97F4A7C15ULL) >> 48) & (HS_CAP - 1);
    while (a->hs[h]) { if (a->hs[h] == key) return 0; h = (h + 1) & (HS_CAP - 1); }
    a->hs[h] = key; a->hs_used[a->hs_n++] = h;
    return 1;
}
static void hs_clear(bso_aligner *a) { for (uint32_t i = 0; i < a->hs_n; i++) a->hs[a->hs_used[i]] = 0; a->hs_n = 0; }

uint32_t bso_myrand(const bso_params *p, uint32_t index, uint32_t *rseed) /* utilities.cpp:40-50 */
{
    if (p->randseed == 0) { *rseed = *rseed * 1103515245u + 12345u; return (*rseed >> 16) & 0x7fff; } /* rand_r: nondeterministic in the reference */
    uint64_t v = ((uint64_t)(int64_t)(int)index + (uint64_t)(int64_t)(int)(p->randseed * 1000000)) * 3935559000370003845ULL + 2691343689449507681ULL;
    v ^= v >> 21; v ^= v << 37; v ^= v >> 4;
    v *= 4768777513237032717ULL;
    v ^= v << 20; v ^= v >> 41; v ^= v << 5;
    return (uint32_t)(v & 0xffffffffULL);
}

/* ------------------------------------------------------------------------------------------------
 * a12: trimming / QC filter
 * ---------------------------------------------------------------------------------------------- */
static void erase_at(bso_aligner *a, int pos) /* seq.erase(pos); if(qual.size()>pos) qual.erase(pos) */
{
    if (a->len > pos) { a->len = pos; a->seq[pos] = 0; }
    if (a->qlen > pos) { a->qlen = pos; a->qual[pos] = 0; }
}

static int trim_adapter(bso_aligner *a) /* SingleAlign::TrimAdapter, align.cpp:371-425 */
{
    const bso_params *p = a->P;
    int i, k, m, m0, pos;
    a->raw_readlen = a->len;
    if (p->rrbs) {
        int dl = (int)strlen(p->digest_site);
        for (i = 0; i < p->n_adapter; i++) {
            int al = (int)strlen(p->adapter[i]);
            for (pos = p->seed_size; pos < a->len - 5; pos++) {
                m0 = 0;
                for (k = 0; k < al && k < 15 && pos + k < a->len; k++)
                    if ((m0 += (p->adapter[i][k] != a->seq[pos + k])) > 4) break;
                if (k < m0 * 5) continue;
                m = m0;
                for (int t = 0; t < dl - p->digest_pos; t++) {
                    char an = p->digest_site[t], rn = a->seq[pos - dl + p->digest_pos + t];
                    m += (an != rn) && (an != 'C' || rn != 'T');
                }
                if (k >= m * 5) { erase_at(a, pos); return 1; }
                if (p->pairend) {
                    m = m0;
                    for (int t = 0; t < dl - p->digest_pos; t++) {
                        char an = p->digest_site[t], rn = a->seq[pos - dl + p->digest_pos + t];
                        m += (an != rn) && (an != 'G' || rn != 'A');
                    }
                    if (k >= m * 5) { erase_at(a, pos); return 1; }
                }
            }
        }
    } else {
        for (i = 0; i < p->n_adapter; i++) {
            int al = (int)strlen(p->adapter[i]);
            for (pos = p->seed_size; pos < a->len - 4; pos++) {
                m0 = 0;
                for (k = 0; k < al && k < 15 && pos + k < a->len; k++)
                    if ((m0 += (p->adapter[i][k] != a->seq[pos + k])) > 4) break;
                if (k >= m0 * 5 && k > 3) { erase_at(a, pos); return 1; }
            }
        }
    }
    return 0;
}

static int trim_low_qual(bso_aligner *a) /* SingleAlign::TrimLowQual, align.cpp:59-79 */
{
    const bso_params *p = a->P;
    if (p->qual_threshold == 0 || a->qlen == 1) return 1;
    int read_zero_qual = (uint8_t)p->zero_qual;
    if (p->out_sam && read_zero_qual != '!') {
        for (int i = 0; i < a->qlen; i++) a->qual[i] = (char)(a->qual[i] - (read_zero_qual - '!'));
        read_zero_qual = '!';
    }
    for (int i = a->qlen; i > 0; i--)
        if ((int)(signed char)a->qual[i - 1] > read_zero_qual + p->qual_threshold) {
            if (i >= p->seed_size) { if (a->qlen > i) { a->qlen = i; a->qual[i] = 0; } if (a->len > i) { a->len = i; a->seq[i] = 0; } return 1; }
        }
    return 0;
}

static int filter_reads(bso_aligner *a) /* SingleAlign::FilterReads, align.cpp:579-589 */
{
    const bso_params *p = a->P;
    trim_adapter(a);
    if (trim_low_qual(a) == 0) return 1;
    if (a->len < p->seed_size) return 1;
    int ns = 0;
    for (int i = 0; i < a->len; i++) if (!p->reg_alphabet[(unsigned char)a->seq[i]]) ns++; /* CountNs, align.cpp:48-55 */
    if (ns > p->max_ns) return 1;
    a->read_max_snp_num = (int)((uint64_t)(p->max_snp_num + 1) * (uint64_t)(a->len - 1) / (uint64_t)a->raw_readlen);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * a7: read packer
 * ---------------------------------------------------------------------------------------------- */
static inline void right_shift(const uint32_t *o, uint32_t *n) /* RightShiftBinSeq, align.cpp:82-87 */
{
    n[0] = o[0] >> 2;
    for (int i = 1; i < BSO_FIXELEMENT; i++) n[i] = (o[i] >> 2) | (o[i - 1] << 30);
}

static void convert_binary_seq(bso_aligner *a) /* SingleAlign::ConvertBinaySeq, align.cpp:90-162 */
{
    const bso_params *p = a->P;
    int i, h; uint32_t _a, _b, s = 0;
    a->flag_chain = p->chains || (a->readset < 2);
    a->cflag_chain = p->chains || (a->readset == 2);
    if (a->flag_chain) {
        h = 0; _a = _b = 0;
        for (i = 1; i <= a->len; i++) {
            unsigned char c = (unsigned char)a->seq[i - 1];
            _a <<= 2; _b <<= 2;
            _a |= p->alphabet[c]; _b |= p->reg_alphabet[c];
            if (i > p->seed_size) { s <<= 2; s |= _a & 3; a->seed_array[i - p->seed_size] = XT(s & p->seed_bits); }
            else if (i == p->seed_size) { s = _a; a->seed_array[0] = XT(s); }
            if (0 == i % SEGLEN) { a->bseq[0][h] = _a; a->reg[0][h++] = _b; _a = _b = 0; }
        }
        for (; i != FIXSIZE + 1; i++) {
            _a <<= 2; _b <<= 2;
            if (0 == i % SEGLEN) { a->bseq[0][h] = _a; a->reg[0][h++] = _b; _a = _b = 0; }
        }
        for (i = 1; i != SEGLEN; i++) { right_shift(a->bseq[i - 1], a->bseq[i]); right_shift(a->reg[i - 1], a->reg[i]); }
        a->n_orient++;
    }
    if (a->cflag_chain) {
        h = 0; _a = _b = 0;
        for (i = 1; i <= a->len; i++) {
            unsigned char c = (unsigned char)a->seq[a->len - i];
            _a <<= 2; _b <<= 2;
            _a |= p->rev_alphabet[c]; _b |= p->reg_alphabet[c];
            if (i > p->seed_size) { s <<= 2; s |= _a & 3; a->cseed_array[i - p->seed_size] = XT(s & p->seed_bits); }
            else if (i == p->seed_size) { s = _a; a->cseed_array[0] = XT(s); }
            if (0 == i % SEGLEN) { a->cbseq[0][h] = _a; a->creg[0][h++] = _b; _a = _b = 0; }
        }
        for (; i != FIXSIZE + 1; i++) {
            _a <<= 2; _b <<= 2;
            if (0 == i % SEGLEN) { a->cbseq[0][h] = _a; a->creg[0][h++] = _b; _a = _b = 0; }
        }
        for (i = 1; i != SEGLEN; i++) { right_shift(a->cbseq[i - 1], a->cbseq[i]); right_shift(a->creg[i - 1], a->creg[i]); }
        a->n_orient++;
    }
}

/* ------------------------------------------------------------------------------------------------
 * a8: seed planner
 * ---------------------------------------------------------------------------------------------- */
static inline uint32_t bucket_hdr0(const bso_ref *r, uint32_t key) /* index2[s]==NULL ? 0 : index2[s][0] (=2+N) */
{
    uint32_t n = r->bucket_off[key + 1] - r->bucket_off[key];
    return n ? n + 2 : 0;
}
static int count_seeds(bso_aligner *a, const uint32_t *arr, int n, int start) /* CountSeeds/CountCSeeds, align.cpp:549-565 */
{
    const bso_params *p = a->P;
    int total = 0;
    for (int i = 0; i < p->index_interval; i++) {
        uint32_t s = arr[p->profile_a[n][i] + start - i];
        total += (int)bucket_hdr0(a->R, s);
    }
    a->n_lookup += p->index_interval;
    return total;
}
static uint32_t total_seed_loc(bso_aligner *a, const uint32_t *arr, int start) /* GetTotalSeedLoc, align.cpp:567-577 */
{
    int total = 0;
    for (int i = 0; i < a->seedseg_num; i++) total += count_seeds(a, arr, i, start);
    return (uint32_t)total;
}
static void adjust_start_array(bso_aligner *a, const uint32_t *arr, int *ssa, int offset) /* AdjustSeedStartArray, align.cpp:506-547 */
{
    const bso_params *p = a->P;
    int i, ptr, start, end, max_offset;
    for (i = 0; i < a->seedseg_num; i++) ssa[i] = offset;
    if (p->rrbs) return;
    max_offset = (a->len - p->index_interval + 1) % p->seed_size;
    for (i = 0; i < a->seedseg_num; i++) {
        if (i % 2 == 0) ptr = i / 2; else ptr = a->seedseg_num - 1 - i / 2;
        uint32_t total = 0xffffffffu;
        start = (ptr == 0) ? 0 : ssa[ptr - 1];
        end = (ptr == a->seedseg_num - 1) ? max_offset : ssa[ptr + 1];
        ssa[ptr] = start;
        for (int ii = start; ii <= end; ii++) {
            uint32_t tt = (uint32_t)count_seeds(a, arr, ptr, ii);
            if (tt < total) { total = tt; ssa[ptr] = ii; }
        }
    }
}
static int seedidx_cmp(const void *x, const void *y)
{
    const uint32_t *a = x, *b = y; /* pair<int,int> operator<; counts are compared as int */
    if ((int)a[0] != (int)b[0]) return (int)a[0] < (int)b[0] ? -1 : 1;
    return (int)a[1] - (int)b[1];
}
static void reorder_seed(bso_aligner *a) /* SingleAlign::ReorderSeed, align.cpp:454-504 */
{
    const bso_params *p = a->P;
    const bso_ref *r = a->R;
    uint32_t i, ii, s, total = 0xffffffffu, ctotal = 0xffffffffu, tt;
    if (p->rrbs) a->seed_start_offset = a->cseed_start_offset = 0;
    else {
        ii = (uint32_t)(a->len - p->index_interval + 1) % p->seed_size;
        for (i = 0; i < ii; i++) {
            if (a->flag_chain) { tt = total_seed_loc(a, a->seed_array, (int)i); if (tt < total) { total = tt; a->seed_start_offset = (int)i; } }
            if (a->cflag_chain) { tt = total_seed_loc(a, a->cseed_array, (int)i); if (tt < ctotal) { ctotal = tt; a->cseed_start_offset = (int)i; } }
        }
    }
    if (a->flag_chain) {
        adjust_start_array(a, a->seed_array, a->seed_start_array, a->seed_start_offset);
        for (i = 0; i < (uint32_t)a->seedseg_num; i++) {
            s = 0;
            if (p->rrbs) { /* GenerateSeeds, align.h:138-150 */
                a->seeds[i][0] = a->seed_array[p->profile_a[i][0] + a->seed_start_array[i]];
                s += r->bucket_off[a->seeds[i][0] + 1] - r->bucket_off[a->seeds[i][0]];
                a->n_lookup++;
            } else {
                for (ii = 0; ii < (uint32_t)p->index_interval; ii++) a->seeds[i][ii] = a->seed_array[p->profile_a[i][ii] + a->seed_start_array[i] - (int)ii];
                for (ii = 0; ii != (uint32_t)p->index_interval; ii++) s += bucket_hdr0(r, a->seeds[i][ii]);
                a->n_lookup += p->index_interval;
            }
            a->seedindex[i].cnt = s; a->seedindex[i].idx = (int)i;
        }
        qsort(a->seedindex, a->seedseg_num, sizeof(a->seedindex[0]), seedidx_cmp);
    }
    if (a->cflag_chain) {
        adjust_start_array(a, a->cseed_array, a->cseed_start_array, a->cseed_start_offset);
        for (i = 0; i < (uint32_t)a->seedseg_num; i++) {
            s = 0;
            if (p->rrbs) { /* GenerateCSeeds, align.h:152-164 */
                a->cseeds[i][0] = a->cseed_array[p->profile_a[i][0] + a->cseed_offset + a->cseed_start_array[i]];
                s += r->bucket_off[a->cseeds[i][0] + 1] - r->bucket_off[a->cseeds[i][0]];
                a->n_lookup++;
            } else {
                for (ii = 0; ii < (uint32_t)p->index_interval; ii++) a->cseeds[i][ii] = a->cseed_array[p->profile_a[i][ii] + a->cseed_start_array[i] - (int)ii];
                for (ii = 0; ii != (uint32_t)p->index_interval; ii++) s += bucket_hdr0(r, a->cseeds[i][ii]);
                a->n_lookup += p->index_interval;
            }
            a->cseedindex[i].cnt = s; a->cseedindex[i].idx = (int)i;
        }
        qsort(a->cseedindex, a->seedseg_num, sizeof(a->cseedindex[0]), seedidx_cmp);
    }
}

/* ------------------------------------------------------------------------------------------------
 * a9: candidate scan + mismatch extension
 * ---------------------------------------------------------------------------------------------- */
static inline uint64_t ld64(const uint32_t *w) { return (uint64_t)w[0] | ((uint64_t)w[1] << 32); }

/* SingleAlign::CountMismatch (READ_144), align.h:187-199; also accounts the 64-bit words touched */
static inline uint32_t count_mismatch(bso_aligner *a, const uint32_t *q, const uint32_t *r, const uint32_t *s)
{
    uint32_t t;
    a->n_cand++;
    uint64_t s0 = ld64(s);
    if ((t = XM64(((ld64(q) & XC64(s0)) ^ s0) & ld64(r))) > a->snp_thres) { a->sum_w += 1; return t; }
    uint64_t s1 = ld64(s + 2);
    if ((t += XM64(((ld64(q + 2) & XC64(s1)) ^ s1) & ld64(r + 2))) > a->snp_thres) { a->sum_w += 2; return t; }
    a->sum_w += 5;
    uint64_t s2 = ld64(s + 4), s3 = ld64(s + 6), s4 = ld64(s + 8);
    return t + XM64(((ld64(q + 4) & XC64(s2)) ^ s2) & ld64(r + 4)) + XM64(((ld64(q + 6) & XC64(s3)) ^ s3) & ld64(r + 6)) +
           XM64(((ld64(q + 8) & XC64(s4)) ^ s4) & ld64(r + 8));
}

static inline void int2hit(const bso_ref *r, uint32_t p, int c, bso_hit *h) /* RefSeq::int2hit, dbseq.cpp:585-595 */
{
    int left = 0, right = (int)r->n_chr, mid;
    while (left < right - 1) { mid = (left + right) / 2; if (p >= r->anchor[mid]) left = mid; else right = mid; }
    h->chr = (uint32_t)left * 2 + c; h->loc = p - r->anchor[left];
}

/* common tail of the four WGBS inner loops (align.cpp:270-278, 287-296, 317-325, 332-341).
 * returns 1 when SnpAlign must return */
static inline int accept_hit(bso_aligner *a, int mode, uint32_t w, bso_hit h, int corient)
{
    const bso_params *p = a->P;
    if ((uint64_t)h.loc + (uint64_t)a->len > (uint64_t)a->R->chr_size[h.chr >> 1]) return 0; /* overflow the end of refseq */
    if (!hs_insert(a, h.chr >> 1, h.loc)) return 0; /* hit already exist */
    if (!corient) a->hits[w][a->n_hit[w]++] = h; else a->chits[w][a->n_chit[w]++] = h;
    if (w == (uint32_t)mode && !p->pairend && p->report_repeat_hits == 0) if (a->n_hit[w] + a->n_chit[w] > 1) return 1;
    if (a->n_hit[w] + a->n_chit[w] >= p->max_num_hits) { if (w == 0) return 1; else a->snp_thres = w - 1; }
    return 0;
}

static void snp_align(bso_aligner *a, int mode) /* SingleAlign::SnpAlign, align.cpp:168-347 */
{
    const bso_params *p = a->P;
    const bso_ref *r = a->R;
    uint32_t i, j, m, w; int h, modeindex;
    bso_hit hit;
    if (p->rrbs) { /* align.cpp:175-252 */
        if (a->flag_chain) {
            modeindex = a->seedindex[mode].idx;
            uint32_t seed = a->seeds[modeindex][0];
            m = r->bucket_off[seed + 1] - r->bucket_off[seed];
            const uint32_t *loc1 = r->rrbs_entries + 2 * (size_t)r->bucket_off[seed];
            h = p->profile_a[modeindex][0];
            for (j = 0; j != m; j++) {
                hit.chr = loc1[2 * j]; hit.loc = loc1[2 * j + 1];
                if ((hit.chr >> 16) != (uint32_t)modeindex) continue;
                hit.chr &= 0xffff;
                if (hit.loc < (uint32_t)h) continue;
                hit.loc -= h;
                uint32_t z = hit.loc % SEGLEN;
                const uint32_t *s = ((hit.chr & 1) ? r->crefcat : r->refcat) + r->anchor[hit.chr / 2] / SEGLEN + hit.loc / SEGLEN;
                w = count_mismatch(a, a->bseq[z], a->reg[z], s);
                if (w > a->snp_thres) continue;
                if (hit.chr & 1) hit.loc = r->rc_offset[hit.chr >> 1] - a->len - hit.loc;
                if ((uint64_t)hit.loc + a->len > r->chr_size[hit.chr >> 1]) continue;
                if (!hs_insert(a, hit.chr >> 1, hit.loc)) continue;
                if (!p->pairend) {
                    uint32_t f; int sl; ccgg_seglen(p, r, hit.chr, hit.loc, a->len, &f, &sl);
                    if (sl > p->max_insert) continue;
                    if (sl < p->min_insert) continue;
                }
                a->hits[w][a->n_hit[w]++] = hit;
                if (w == (uint32_t)mode && !p->pairend && p->report_repeat_hits == 0) if (a->n_hit[w] + a->n_chit[w] > 1) return;
                if (a->n_hit[w] + a->n_chit[w] >= p->max_num_hits) { if (w == 0) return; else a->snp_thres = w - 1; }
            }
        }
        if (a->cflag_chain) {
            modeindex = a->cseedindex[mode].idx;
            int cmodeindex = a->len / p->seed_size - 1 - modeindex;
            uint32_t seed = a->cseeds[modeindex][0];
            m = r->bucket_off[seed + 1] - r->bucket_off[seed];
            const uint32_t *loc1 = r->rrbs_entries + 2 * (size_t)r->bucket_off[seed];
            h = p->profile_a[modeindex][0] + (int)a->cseed_offset;
            for (j = 0; j != m; j++) {
                hit.chr = loc1[2 * j]; hit.loc = loc1[2 * j + 1];
                if (((hit.chr ^ 0x1000000u) >> 16) != (uint32_t)cmodeindex) continue;
                hit.chr &= 0xffff;
                if (hit.loc < (uint32_t)h) continue;
                hit.loc -= h;
                uint32_t z = hit.loc % SEGLEN;
                const uint32_t *s = ((hit.chr & 1) ? r->crefcat : r->refcat) + r->anchor[hit.chr / 2] / SEGLEN + hit.loc / SEGLEN;
                w = count_mismatch(a, a->cbseq[z], a->creg[z], s);
                if (w > a->snp_thres) continue;
                if (hit.chr & 1) hit.loc = r->rc_offset[hit.chr >> 1] - a->len - hit.loc;
                if ((uint64_t)hit.loc + a->len > r->chr_size[hit.chr >> 1]) continue;
                if (!hs_insert(a, hit.chr >> 1, hit.loc)) continue;
                a->chits[w][a->n_chit[w]++] = hit;
                if (w == (uint32_t)mode && !p->pairend && p->report_repeat_hits == 0) if (a->n_hit[w] + a->n_chit[w] > 1) return;
                if (a->n_hit[w] + a->n_chit[w] >= p->max_num_hits) { if (w == 0) return; else a->snp_thres = w - 1; }
            }
        }
        return;
    }
    for (int orient = 0; orient < 2; orient++) { /* direct chain (align.cpp:255-300) then complementary chain (:302-345) */
        if (orient == 0 ? !a->flag_chain : !a->cflag_chain) continue;
        modeindex = orient ? a->cseedindex[mode].idx : a->seedindex[mode].idx;
        uint32_t(*bs)[10] = orient ? a->cbseq : a->bseq, (*rg)[10] = orient ? a->creg : a->reg;
        for (i = 0; i != (uint32_t)p->index_interval; i++) {
            uint32_t seed = orient ? a->cseeds[modeindex][i] : a->seeds[modeindex][i];
            uint32_t b0 = r->bucket_off[seed], b1 = r->bucket_off[seed + 1];
            if (b0 == b1) continue; /* index2[_seed]==NULL */
            uint32_t mc = b0 + r->bucket_nfwd[seed];
            h = -(int)p->profile_a[modeindex][i] + (int)i - (orient ? a->cseed_start_array[modeindex] : a->seed_start_array[modeindex]);
            for (j = b0; j != mc; j++) {
                uint32_t loc = r->entries[j] + (uint32_t)h;
                w = count_mismatch(a, bs[loc % SEGLEN], rg[loc % SEGLEN], r->refcat + loc / SEGLEN);
                if (w > a->snp_thres) continue;
                int2hit(r, loc, 0, &hit);
                if (accept_hit(a, mode, w, hit, orient)) return;
            }
            for (; j != b1; j++) {
                uint32_t loc = r->entries[j] + (uint32_t)h;
                w = count_mismatch(a, bs[loc % SEGLEN], rg[loc % SEGLEN], r->crefcat + loc / SEGLEN);
                if (w > a->snp_thres) continue;
                int2hit(r, loc, 1, &hit);
                hit.loc = r->rc_offset[hit.chr >> 1] - (uint32_t)a->len - hit.loc;
                if (accept_hit(a, mode, w, hit, orient)) return;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * a10: per-read driver
 * ---------------------------------------------------------------------------------------------- */
static void clear_hits(bso_aligner *a) /* ClearHits, align.cpp:428-433 */
{
    for (int i = 0; i <= a->P->max_snp_num; i++) a->n_hit[i] = a->n_chit[i] = 0;
    hs_clear(a);
}
static void reset_leak_state(bso_aligner *a)
{
    if (a->leak_mode) return;
    a->seed_start_offset = a->cseed_start_offset = 0;
    memset(a->seed_array, 0, sizeof(a->seed_array)); memset(a->cseed_array, 0, sizeof(a->cseed_array));
}
static void prepare_align(bso_aligner *a) /* common head of SingleAlign::RunAlign (:438-443) and PairAlign::RunAlign (:144-160) */
{
    const bso_params *p = a->P;
    int x = (a->len - p->index_interval + 1) / p->seed_size, y = a->read_max_snp_num + 1;
    a->seedseg_num = x < y ? x : y;
}
static int run_align(bso_aligner *a) /* SingleAlign::RunAlign, align.cpp:435-452 */
{
    const bso_params *p = a->P;
    int i;
    clear_hits(a);
    reset_leak_state(a);
    prepare_align(a);
    convert_binary_seq(a);
    a->snp_thres = (uint32_t)a->read_max_snp_num;
    a->cseed_offset = (uint32_t)a->len % p->seed_size;
    reorder_seed(a);
    for (i = 0; i < a->seedseg_num; i++) {
        snp_align(a, i);
        if (!p->rrbs) for (int ii = 0; ii <= i; ii++) if (a->n_hit[ii] || a->n_chit[ii]) return 1;
    }
    for (i = 0; i <= a->read_max_snp_num; i++) if (a->n_hit[i] || a->n_chit[i]) return 1;
    return 0;
}

static void set_read(bso_aligner *a, uint32_t index, int readset, const char *seq, const char *qual)
{
    const bso_params *p = a->P;
    int n = (int)strlen(seq);
    if (n > p->max_readlen) n = p->max_readlen; /* reads.cpp:115-117 */
    if (n > FIXSIZE) n = FIXSIZE;
    memcpy(a->seq, seq, n); a->seq[n] = 0; a->len = n;
    if (qual) {
        /* the quality string keeps its own length (a malformed record may carry more or fewer characters than bases): the reader only cuts it
         * where it cuts the read (reads.cpp:115-117), and TrimLowQual scans the whole of it (align.cpp:69-78) */
        int q = (int)strlen(qual);
        if ((int)strlen(seq) > p->max_readlen && q > p->max_readlen) q = p->max_readlen;
        if (q > FIXSIZE + 63) q = FIXSIZE + 63;
        memcpy(a->qual, qual, q); a->qual[q] = 0; a->qlen = q;
    }
    else { memset(a->qual, p->zero_qual + 40, n); a->qual[n] = 0; a->qlen = n; } /* reads.cpp:108 default_qual */
    a->index = index; a->readset = readset;
}

static void fill_state(const bso_aligner *a, int filtered, bso_read_result *o)
{
    memset(o, 0, sizeof(*o));
    o->filtered = filtered; o->len = a->len; o->raw_len = a->raw_readlen; o->best_class = -1;
    if (filtered) return;
    o->read_max_snp_num = a->read_max_snp_num; o->seedseg_num = a->seedseg_num;
    o->flag_chain = a->flag_chain; o->cflag_chain = a->cflag_chain;
    /* only what this read's ReorderSeed wrote: entries from seedseg_num on (and the arrays of a chain the read does not search)
     * still hold what earlier reads of the same aligner left there - the worker that happened to run them, under threads */
    for (int i = 0; i < a->seedseg_num && i < 16; i++) {
        if (a->flag_chain) { o->seed_start_array[i] = a->seed_start_array[i]; o->seedindex[i] = a->seedindex[i].idx; o->seedcount[i] = a->seedindex[i].cnt; }
        if (a->cflag_chain) { o->cseed_start_array[i] = a->cseed_start_array[i]; o->cseedindex[i] = a->cseedindex[i].idx; o->cseedcount[i] = a->cseedindex[i].cnt; }
    }
    for (int i = 0; i <= a->P->max_snp_num && i < 16; i++) { o->n_hit[i] = a->n_hit[i]; o->n_chit[i] = a->n_chit[i]; }
    o->snp_thres = a->snp_thres;
}

static void select_hit(bso_aligner *a, bso_read_result *o) /* SingleAlign::StringAlign, align.cpp:610-627 */
{
    int ii, sum = 0, j;
    for (ii = 0; ii <= a->read_max_snp_num; ii++) if ((sum = a->n_hit[ii] + a->n_chit[ii]) > 0) break;
    o->n_best = sum;
    if (sum == 0) { o->best_class = -1; return; }
    o->best_class = ii;
    j = (int)(bso_myrand(a->P, a->index, &a->rand_rseed) % (uint32_t)sum);
    if (j < a->n_hit[ii]) { o->chain = 0; o->chr = a->hits[ii][j].chr; o->loc = a->hits[ii][j].loc; }
    else { o->chain = 1; o->chr = a->chits[ii][j - a->n_hit[ii]].chr; o->loc = a->chits[ii][j - a->n_hit[ii]].loc; }
}

int bso_se_align(bso_aligner *a, uint32_t index, int readset, const char *seq, const char *qual, bso_read_result *out)
{
    set_read(a, index, readset, seq, qual);
    int f = filter_reads(a);
    if (!f) run_align(a);
    fill_state(a, f, out);
    if (!f) select_hit(a, out);
    return f;
}

/* ------------------------------------------------------------------------------------------------
 * a11: paired-end
 * ---------------------------------------------------------------------------------------------- */
static int hit_cmp(const void *x, const void *y) /* HitComp, utilities.cpp:53-55 */
{
    const bso_hit *a = x, *b = y;
    if (a->chr != b->chr) return a->chr < b->chr ? -1 : 1;
    if (a->loc != b->loc) return a->loc < b->loc ? -1 : 1;
    return 0;
}
static void sort_hits_4pe(bso_aligner *a, int n) /* SortHits4PE, align.cpp:363-368 */
{
    qsort(a->hits[n], a->n_hit[n], sizeof(bso_hit), hit_cmp);
    qsort(a->chits[n], a->n_chit[n], sizeof(bso_hit), hit_cmp);
}

static int get_pairs(bso_aligner *sa, bso_aligner *sb, int na, int nb) /* PairAlign::GetPairs, pairs.cpp:34-135 */
{
    const bso_params *p = sa->P;
    int i, j, insert_size, bstart, bend;
    uint32_t seg_start, seg_end, chra;
    if (na > sa->read_max_snp_num || nb > sb->read_max_snp_num) return 0;
    bso_pair pp; memset(&pp, 0, sizeof(pp));
    pp.na = (uint8_t)na; pp.nb = (uint8_t)nb;
    uint32_t *cnt = &sa->n_pairs[na + nb];
    /* a+ vs b- */
    pp.chain = 0; chra = ~0u; bstart = 0; bend = 0;
    for (i = 0; i < sa->n_hit[na]; i++) {
        if (chra != sa->hits[na][i].chr) {
            chra = sa->hits[na][i].chr;
            for (bstart = bend; bstart < sb->n_chit[nb]; bstart++) if (sb->chits[nb][bstart].chr >= chra) break;
            for (bend = bstart; bend < sb->n_chit[nb]; bend++) if (sb->chits[nb][bend].chr > chra) break;
        }
        for (j = bstart; j < bend; j++) {
            if (chra & 1) { seg_start = sb->chits[nb][j].loc; seg_end = sa->hits[na][i].loc + (uint32_t)sa->len; }
            else { seg_start = sa->hits[na][i].loc; seg_end = sb->chits[nb][j].loc + (uint32_t)sb->len; }
            insert_size = (int)(seg_end - seg_start);
            if (insert_size >= p->min_insert && insert_size <= p->max_insert) {
                pp.a = sa->hits[na][i]; pp.b = sb->chits[nb][j]; pp.insert = insert_size;
                sa->pairhits[na + nb][(*cnt)++] = pp;
                if (*cnt >= (uint32_t)p->max_num_hits) return 1;
            }
        }
    }
    /* a- vs b+ */
    pp.chain = 1; chra = ~0u; bstart = 0; bend = 0;
    for (i = 0; i < sa->n_chit[na]; i++) {
        if (chra != sa->chits[na][i].chr) {
            chra = sa->chits[na][i].chr;
            for (bstart = bend; bstart < sb->n_hit[nb]; bstart++) if (sb->hits[nb][bstart].chr >= chra) break;
            for (bend = bstart; bend < sb->n_hit[nb]; bend++) if (sb->hits[nb][bend].chr > chra) break;
        }
        for (j = bstart; j < bend; j++) {
            if ((chra & 1) == 0) { seg_start = sb->hits[nb][j].loc; seg_end = sa->chits[na][i].loc + (uint32_t)sa->len; }
            else { seg_start = sa->chits[na][i].loc; seg_end = sb->hits[nb][j].loc + (uint32_t)sb->len; }
            insert_size = (int)(seg_end - seg_start);
            if (insert_size >= p->min_insert && insert_size <= p->max_insert) {
                pp.a = sa->chits[na][i]; pp.b = sb->hits[nb][j]; pp.insert = insert_size;
                sa->pairhits[na + nb][(*cnt)++] = pp;
                if (*cnt >= (uint32_t)p->max_num_hits) return 1;
            }
        }
    }
    if (*cnt > 0) return 1;
    return 0;
}

static int pair_run_align(bso_aligner *sa, bso_aligner *sb) /* PairAlign::RunAlign, pairs.cpp:137-190 */
{
    const bso_params *p = sa->P;
    int n, i, j;
    for (i = 0; i <= p->max_snp_num * 2; i++) sa->n_pairs[i] = 0;
    clear_hits(sa); clear_hits(sb);
    reset_leak_state(sa); reset_leak_state(sb);
    prepare_align(sa); prepare_align(sb);
    convert_binary_seq(sa); convert_binary_seq(sb);
    sa->snp_thres = (uint32_t)sa->read_max_snp_num; sb->snp_thres = (uint32_t)sb->read_max_snp_num;
    sa->cseed_offset = (uint32_t)sa->len % p->seed_size; sb->cseed_offset = (uint32_t)sb->len % p->seed_size;
    reorder_seed(sa); reorder_seed(sb);
    int maxi = sa->read_max_snp_num > sb->read_max_snp_num ? sa->read_max_snp_num : sb->read_max_snp_num;
    for (i = 0; i <= maxi; i++) {
        if (i < sa->seedseg_num) snp_align(sa, i);
        if (i < sb->seedseg_num) snp_align(sb, i);
        if (i <= sa->read_max_snp_num) sort_hits_4pe(sa, i);
        if (i <= sb->read_max_snp_num) sort_hits_4pe(sb, i);
        n = get_pairs(sa, sb, i, i);
        for (j = 0; j < i; j++) n += get_pairs(sa, sb, i, j) + get_pairs(sa, sb, j, i);
        if (n > 0) return i + 1;
    }
    return 0;
}

static void fix_unpaired_short_fragment(bso_aligner *a) /* align.cpp:768-791 */
{
    const bso_params *p = a->P;
    int ii, j, k; uint32_t f; int sl;
    if (a->len >= p->min_insert) return;
    for (ii = 0; ii <= a->read_max_snp_num; ii++) {
        for (j = 0; j < a->n_hit[ii]; j++) {
            ccgg_seglen(p, a->R, a->hits[ii][j].chr, a->hits[ii][j].loc, a->len, &f, &sl);
            if (sl < p->min_insert || sl > p->max_insert) { a->n_hit[ii]--; for (k = j; k < a->n_hit[ii]; k++) a->hits[ii][k] = a->hits[ii][k + 1]; j--; }
        }
        for (j = 0; j < a->n_chit[ii]; j++) {
            ccgg_seglen(p, a->R, a->chits[ii][j].chr, a->chits[ii][j].loc, a->len, &f, &sl);
            if (sl < p->min_insert || sl > p->max_insert) { a->n_chit[ii]--; for (k = j; k < a->n_chit[ii]; k++) a->chits[ii][k] = a->chits[ii][k + 1]; j--; }
        }
        if (a->n_hit[ii] + a->n_chit[ii] > 0) break;
    }
}

static void select_unpair(bso_aligner *a, int filtered, bso_read_result *o) /* one half of StringAlignUnpair, pairs.cpp:255-275 */
{
    int na = 0, ma = -1, ra = 0;
    o->best_class = -1; o->n_best = -1; o->chain = 0;
    if (!filtered) {
        for (na = 0; na <= a->read_max_snp_num; na++) if ((ma = a->n_hit[na] + a->n_chit[na]) > 0) break;
        if (ma) {
            if (ma > 1) ra = (int)(bso_myrand(a->P, a->index, &a->rand_rseed) % (uint32_t)ma);
            bso_hit h = (ra < a->n_hit[na]) ? a->hits[na][ra] : a->chits[na][ra - a->n_hit[na]];
            o->chr = h.chr; o->loc = h.loc;
        }
        na %= (a->read_max_snp_num + 1);
        o->chain = (ra >= a->n_hit[na]);
        o->best_class = na;
    }
    o->n_best = ma;
}

int bso_pe_align(bso_aligner *sa, bso_aligner *sb, uint32_t index, const char *seq_a, const char *qual_a,
                 const char *seq_b, const char *qual_b, bso_pair_result *out)
{
    const bso_params *p = sa->P;
    set_read(sa, index, 1, seq_a, qual_a);
    set_read(sb, index, 2, seq_b, qual_b);
    memset(out, 0, sizeof(*out));
    int f1 = filter_reads(sa), f2 = filter_reads(sb), paired, tmp = 0;
    for (int i = 0; i <= 2 * BSO_MAXSNPS; i++) sa->n_pairs[i] = 0;
    if (f1 == 0 && f2 == 0) paired = pair_run_align(sa, sb);
    else { paired = 0; if (f1 == 0) run_align(sa); if (f2 == 0) run_align(sb); }
    fill_state(sa, f1, &out->a); fill_state(sb, f2, &out->b);
    out->paired = paired; out->pair_class = -1;
    if (f1 == 0 && f2 == 0) for (int i = 0; i <= p->max_snp_num * 2; i++) out->n_pairs[i] = sa->n_pairs[i];
    if (paired) { /* StringAlignPair, pairs.cpp:222-242 */
        tmp = 1;
        for (int i = 0; i <= p->max_snp_num * 2; i++) {
            if (0 == sa->n_pairs[i]) continue;
            out->pair_class = i; out->pair_n = (int)sa->n_pairs[i];
            if (1 == sa->n_pairs[i]) { out->pick = sa->pairhits[i][0]; tmp = 0; }
            else if (1 == p->report_repeat_hits) { int j = (int)(bso_myrand(p, sa->index, &sa->rand_rseed) % sa->n_pairs[i]); out->pick = sa->pairhits[i][j]; tmp = 0; }
            break;
        }
    }
    out->tmp = tmp;
    if (tmp == 1 || paired == 0) { /* StringAlignUnpair, pairs.cpp:244-286 */
        if (p->rrbs) { if (!f1) fix_unpaired_short_fragment(sa); if (!f2) fix_unpaired_short_fragment(sb); }
        select_unpair(sa, f1, &out->a);
        select_unpair(sb, f2, &out->b);
        /* counts may have been edited by Fix_Unpaired_Short_Fragment */
        if (p->rrbs) for (int i = 0; i <= p->max_snp_num && i < 16; i++) {
            if (!f1) { out->a.n_hit[i] = sa->n_hit[i]; out->a.n_chit[i] = sa->n_chit[i]; }
            if (!f2) { out->b.n_hit[i] = sb->n_hit[i]; out->b.n_chit[i] = sb->n_chit[i]; }
        }
    }
    return paired;
}

/* ------------------------------------------------------------------------------------------------
 * batch helpers (threading model of main.cpp:49-73: workers pull fixed-size chunks of reads)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint8_t filtered, len; } read_meta;
typedef struct {
    const bso_params *p; const bso_ref *r; uint32_t n; const char *sa; const uint64_t *oa; const char *qa;
    const char *sb; const uint64_t *ob; const char *qb; uint32_t first; void *res; int pe;
    volatile uint32_t *next; uint32_t chunk; uint64_t cnt[4];
    /* leak_mode 1 ("-p 1 exact"): the reference's planner state (align.h:82-91) runs through the whole input in order.  The
     * batch driver still works in parallel chunks: before a chunk, its worker re-establishes the state the sequential run has
     * at the chunk's first read by replaying the PLANNER ONLY (FilterReads, ConvertBinaySeq, ReorderSeed - no SnpAlign) from a
     * read that overwrites everything earlier reads left behind: the nearest unfiltered read before the chunk that sets the
     * start offset ((len - I + 1) % S != 0) and is as long as any unfiltered read before the chunk (entries beyond its last hash
     * were then never written: zero, as in a fresh object).  meta[mate][i] = FilterReads' verdict and trimmed length of read i,
     * pmax[mate][i] = longest unfiltered read among reads [0, i).  tests/test_oracle_golden.py checks this driver against one
     * aligner object fed in order. */
    int leak_mode; const read_meta *meta[2]; const uint8_t *pmax[2];
} batch_job;
/* units per grab: fine enough that every thread gets work even on a bounded sample (the reference's threads grab 50 000
 * reads at a time, main.cpp:49-73 — far coarser) */

static void plan_only(bso_aligner *a, uint32_t index, int readset, const char *seq, const char *qual)
{
    set_read(a, index, readset, seq, qual);
    if (filter_reads(a)) return;              /* RunAlign is not called for a rejected read: state untouched (align.cpp:598) */
    prepare_align(a);
    convert_binary_seq(a);
    a->snp_thres = (uint32_t)a->read_max_snp_num;
    a->cseed_offset = (uint32_t)a->len % a->P->seed_size;
    reorder_seed(a);
}

static void fetch_read(const char *s, const uint64_t *o, const char *q, uint32_t i, char *sbuf, char *qbuf)
{
    uint64_t l = o[i + 1] - o[i]; if (l > FIXSIZE) l = FIXSIZE;
    memcpy(sbuf, s + o[i], l); sbuf[l] = 0;
    if (q) { memcpy(qbuf, q + o[i], l); qbuf[l] = 0; }
}

/* first read of the planner-only replay that gives mate stream m its sequential state at read lo */
static uint32_t replay_start(const batch_job *jb, int m, uint32_t lo)
{
    const int S = jb->p->seed_size, I = jb->p->index_interval;
    const uint8_t top = jb->pmax[m][lo];
    if (top == 0) return lo;                  /* no unfiltered read before lo: the state is still the initial one */
    for (uint32_t i = lo; i-- > 0;) {
        const read_meta t = jb->meta[m][i];
        if (!t.filtered && t.len == top && (t.len - I + 1) % S != 0) return i;
    }
    return 0;                                 /* the longest reads never set the offset: replay the whole prefix */
}

static void *batch_worker(void *arg)
{
    batch_job *jb = arg;
    bso_aligner *a = bso_aligner_new(jb->p, jb->r, jb->leak_mode), *b = jb->pe ? bso_aligner_new(jb->p, jb->r, jb->leak_mode) : NULL;
    char s1[FIXSIZE + 64], q1[FIXSIZE + 64], s2[FIXSIZE + 64], q2[FIXSIZE + 64];
    while (1) {
        uint32_t lo = __sync_fetch_and_add(jb->next, jb->chunk);
        if (lo >= jb->n) break;
        uint32_t hi = lo + jb->chunk < jb->n ? lo + jb->chunk : jb->n;
        if (jb->leak_mode) {
            bso_aligner *al[2] = {a, b};
            for (int m = 0; m < (jb->pe ? 2 : 1); m++) {
                bso_aligner *x = al[m];
                const uint64_t keep[4] = {x->n_lookup, x->n_cand, x->sum_w, x->n_orient};
                x->seed_start_offset = x->cseed_start_offset = 0;
                memset(x->seed_array, 0, sizeof(x->seed_array)); memset(x->cseed_array, 0, sizeof(x->cseed_array));
                for (uint32_t i = replay_start(jb, m, lo); i < lo; i++) {
                    fetch_read(m ? jb->sb : jb->sa, m ? jb->ob : jb->oa, m ? jb->qb : jb->qa, i, s1, q1);
                    plan_only(x, jb->first + i, jb->pe ? m + 1 : 0, s1, (m ? jb->qb : jb->qa) ? q1 : NULL);
                }
                x->n_lookup = keep[0]; x->n_cand = keep[1]; x->sum_w = keep[2]; x->n_orient = keep[3];
            }
        }
        for (uint32_t i = lo; i < hi; i++) {
            fetch_read(jb->sa, jb->oa, jb->qa, i, s1, q1);
            if (!jb->pe) bso_se_align(a, jb->first + i, 0, s1, jb->qa ? q1 : NULL, &((bso_read_result *)jb->res)[i]);
            else {
                fetch_read(jb->sb, jb->ob, jb->qb, i, s2, q2);
                bso_pe_align(a, b, jb->first + i, s1, jb->qa ? q1 : NULL, s2, jb->qb ? q2 : NULL, &((bso_pair_result *)jb->res)[i]);
            }
        }
    }
    jb->cnt[0] = a->n_lookup + (b ? b->n_lookup : 0); jb->cnt[1] = a->n_cand + (b ? b->n_cand : 0);
    jb->cnt[2] = a->sum_w + (b ? b->sum_w : 0); jb->cnt[3] = a->n_orient + (b ? b->n_orient : 0);
    bso_aligner_free(a); bso_aligner_free(b);
    return NULL;
}

static int run_batch(batch_job *proto, int n_threads, uint64_t counters[4])
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 512) n_threads = 512;
    volatile uint32_t next = 0;
    uint32_t chunk = proto->n / ((uint32_t)n_threads * 8u);
    proto->chunk = chunk < 16 ? 16 : chunk > 2048 ? 2048 : chunk;
    read_meta *meta[2] = {NULL, NULL}; uint8_t *pmax[2] = {NULL, NULL};
    if (proto->leak_mode) {   /* FilterReads' verdict and trimmed length of every read (no index access: cheap, sequential) */
        bso_aligner *t = bso_aligner_new(proto->p, proto->r, 0);
        char s[FIXSIZE + 64], q[FIXSIZE + 64];
        for (int m = 0; m < (proto->pe ? 2 : 1); m++) {
            meta[m] = calloc((size_t)proto->n + 1, sizeof(read_meta)); pmax[m] = calloc((size_t)proto->n + 1, 1);
            const char *sq = m ? proto->sb : proto->sa, *ql = m ? proto->qb : proto->qa; const uint64_t *of = m ? proto->ob : proto->oa;
            uint8_t top = 0;
            for (uint32_t i = 0; i < proto->n; i++) {
                pmax[m][i] = top;
                fetch_read(sq, of, ql, i, s, q);
                set_read(t, proto->first + i, proto->pe ? m + 1 : 0, s, ql ? q : NULL);
                meta[m][i].filtered = (uint8_t)filter_reads(t); meta[m][i].len = (uint8_t)t->len;
                if (!meta[m][i].filtered && meta[m][i].len > top) top = meta[m][i].len;
            }
            pmax[m][proto->n] = top;
            proto->meta[m] = meta[m]; proto->pmax[m] = pmax[m];
        }
        bso_aligner_free(t);
    }
    batch_job *jobs = calloc(n_threads, sizeof(batch_job));
    pthread_t *th = calloc(n_threads, sizeof(pthread_t));
    for (int t = 0; t < n_threads; t++) { jobs[t] = *proto; jobs[t].next = &next; }
    for (int t = 1; t < n_threads; t++) pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
    batch_worker(&jobs[0]);
    for (int t = 1; t < n_threads; t++) pthread_join(th[t], NULL);
    if (counters) { memset(counters, 0, 32); for (int t = 0; t < n_threads; t++) for (int k = 0; k < 4; k++) counters[k] += jobs[t].cnt[k]; }
    free(jobs); free(th);
    for (int m = 0; m < 2; m++) { free(meta[m]); free(pmax[m]); }
    return 0;
}

int bso_se_batch_leak(const bso_params *p, const bso_ref *r, uint32_t n_reads, const char *seqs, const uint64_t *off,
                      const char *quals, uint32_t first_index, int n_threads, int leak_mode, bso_read_result *results, uint64_t counters[4])
{
    batch_job j; memset(&j, 0, sizeof(j));
    j.p = p; j.r = r; j.n = n_reads; j.sa = seqs; j.oa = off; j.qa = quals; j.first = first_index; j.res = results; j.pe = 0; j.leak_mode = leak_mode;
    return run_batch(&j, n_threads, counters);
}
int bso_pe_batch_leak(const bso_params *p, const bso_ref *r, uint32_t n_pairs, const char *seqs_a, const uint64_t *off_a,
                      const char *quals_a, const char *seqs_b, const uint64_t *off_b, const char *quals_b,
                      uint32_t first_index, int n_threads, int leak_mode, bso_pair_result *results, uint64_t counters[4])
{
    batch_job j; memset(&j, 0, sizeof(j));
    j.p = p; j.r = r; j.n = n_pairs; j.sa = seqs_a; j.oa = off_a; j.qa = quals_a; j.sb = seqs_b; j.ob = off_b; j.qb = quals_b;
    j.first = first_index; j.res = results; j.pe = 1; j.leak_mode = leak_mode;
    return run_batch(&j, n_threads, counters);
}
int bso_se_batch(const bso_params *p, const bso_ref *r, uint32_t n_reads, const char *seqs, const uint64_t *off,
                 const char *quals, uint32_t first_index, int n_threads, bso_read_result *results, uint64_t counters[4])
{
    return bso_se_batch_leak(p, r, n_reads, seqs, off, quals, first_index, n_threads, 0, results, counters);
}
int bso_pe_batch(const bso_params *p, const bso_ref *r, uint32_t n_pairs, const char *seqs_a, const uint64_t *off_a,
                 const char *quals_a, const char *seqs_b, const uint64_t *off_b, const char *quals_b,
                 uint32_t first_index, int n_threads, bso_pair_result *results, uint64_t counters[4])
{
    return bso_pe_batch_leak(p, r, n_pairs, seqs_a, off_a, quals_a, seqs_b, off_b, quals_b, first_index, n_threads, 0, results, counters);
}
