/* bsx_oracle.h — CPU restatement of the BSMAP v2.6 hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
 * only as the checker / timed CPU baseline; the product (bsmap_amd/, libbsx.so) never links, loads
 * or calls it.  Parity pinning: tests/test_oracle_vs_reference.py checks every function below against
 * the real reference compiled into oracle/_ref/ (white-box harness + the bsmap binary) and against
 * the golden vectors committed under tests/golden/.
 *
 * All file:line citations are relative to /root/reference.
 */
#ifndef BSX_ORACLE_H
#define BSX_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BSO_MAXSNPS 15    /* param.h:27 */
#define BSO_MAXHITS 1000  /* makefile:4 */
#define BSO_FIXELEMENT 10 /* param.h:24 (READ_144) */
#define BSO_REF_MARGIN 400 /* dbseq.h:15 */

typedef struct bso_params {
    /* user options (main.cpp:234-289) */
    int seed_size, index_interval, max_snp_num, max_num_hits, chains, pairend;
    int min_insert, max_insert, report_repeat_hits, randseed;
    int qual_threshold, zero_qual, max_ns, max_readlen, out_sam;
    int rrbs;                 /* RRBS_flag */
    char digest_site[32];     /* without the '-' */
    int digest_pos;
    int n_adapter;
    char adapter[10][128];
    char read_nt, ref_nt;     /* -M, default "TC" */
    /* derived by bso_params_finish() */
    uint8_t alphabet[256], rev_alphabet[256], reg_alphabet[256];
    char useful_nt[9];
    uint8_t profile_a[BSO_MAXSNPS + 1][16]; /* SeedProfile.a, param.cpp:85-93 */
    uint32_t seed_bits;
    int max_seedseg_num;      /* dbseq.cpp:217 */
    uint32_t total_kmers;     /* 3^seed_size */
} bso_params;

void bso_params_default(bso_params *p);              /* Param::Param, param.cpp:6-83 */
int bso_params_set_digest(bso_params *p, const char *site_with_dash); /* param.cpp:95-106 */
int bso_params_finish(bso_params *p);                /* SetAlign + InitMapping */
uint32_t bso_xt(const bso_params *p, uint32_t two_bit_seed); /* Param::XT, param.h:123 */

typedef struct bso_ref {
    uint32_t n_chr;
    uint64_t n_words;          /* words in refcat / crefcat incl. 2*REF_MARGIN */
    uint32_t *refcat, *crefcat;
    uint32_t *anchor;          /* [n_chr+1] global nt coordinate of each chr start */
    uint32_t *chr_size, *rc_offset;
    char **names;
    uint32_t n_blocks;
    uint32_t *blk_id, *blk_begin, *blk_end;   /* sorted by (id, begin), ids 2c (fwd) / 2c+1 (rc) */
    uint64_t sum_length;
    /* WGBS index in CSR form: bucket k = entries[off[k] .. off[k+1]), first nfwd[k] are refcat hits */
    uint32_t total_kmers;
    uint32_t *bucket_off, *bucket_nfwd, *entries;
    uint64_t n_entries;
    /* RRBS: entries are pairs (tag = chr | seg<<16 | dir<<24, loc) in rrbs_entries, same bucket_off */
    uint32_t *rrbs_entries;
    uint32_t **sites; uint32_t *n_sites;  /* CCGG_sites per chr */
} bso_ref;

/* RefSeq::Run_ConvertBinseq, dbseq.cpp:215-282 (text = whole FASTA file contents) */
bso_ref *bso_ref_from_fasta_text(const bso_params *p, const char *text, uint64_t n);
bso_ref *bso_ref_from_fasta_file(const bso_params *p, const char *path);
/* RefSeq::CreateIndex, dbseq.cpp:516-539 */
int bso_index_build(const bso_params *p, bso_ref *r);
/* attach an externally built CSR index (arrays are borrowed, not copied) */
void bso_index_attach(bso_ref *r, uint32_t total_kmers, uint32_t *bucket_off, uint32_t *bucket_nfwd,
                      uint32_t *entries, uint64_t n_entries);
/* wrap externally packed reference arrays (borrowed) */
bso_ref *bso_ref_wrap(uint32_t n_chr, uint64_t n_words, uint32_t *refcat, uint32_t *crefcat,
                      uint32_t *anchor, uint32_t *chr_size, uint32_t *rc_offset);
void bso_ref_free(bso_ref *r);

typedef struct bso_hit { uint32_t chr, loc; } bso_hit;

typedef struct bso_read_result {
    int filtered;          /* FilterReads() != 0 */
    int len, raw_len, read_max_snp_num, seedseg_num;
    int flag_chain, cflag_chain;
    int seed_start_array[16], cseed_start_array[16];
    int seedindex[16], cseedindex[16];
    uint32_t seedcount[16], cseedcount[16];
    int n_hit[16], n_chit[16];
    uint32_t snp_thres;
    /* StringAlign selection, align.cpp:610-627 */
    int best_class;        /* first non-empty class, -1 if none */
    int n_best;            /* hits in that class (fwd + rc orientation) */
    int chain;             /* 0: read as given, 1: reverse-complement orientation */
    uint32_t chr, loc;     /* chr = 2*c (+strand) / 2*c+1 (-strand), loc 0-based forward coordinate */
} bso_read_result;

typedef struct bso_pair { uint16_t chain; uint8_t na, nb; int32_t insert; bso_hit a, b; } bso_pair; /* pairs.h:13-20 */

typedef struct bso_pair_result {
    int paired;            /* PairAlign::RunAlign return */
    int tmp;               /* StringAlignPair return (1 = fall through to unpaired output) */
    uint32_t n_pairs[2 * BSO_MAXSNPS + 1];
    int pair_class;        /* first non-empty pair class or -1 */
    int pair_n;            /* pairs in that class */
    bso_pair pick;         /* chosen pair (valid when tmp==0 && paired) */
    bso_read_result a, b;  /* per-mate state; when unpaired output applies: a/b selections as StringAlignUnpair */
} bso_pair_result;

typedef struct bso_aligner bso_aligner;
/* leak_mode 0: every read starts from zeroed planner state (the product's defined semantics);
 * leak_mode 1: seed_start_offset / seed_array persist between reads exactly as in the reference
 *              object (align.h:82-91) so call-order dependent results can be reproduced. */
bso_aligner *bso_aligner_new(const bso_params *p, const bso_ref *r, int leak_mode);
void bso_aligner_free(bso_aligner *a);
/* body of SingleAlign::Do_Batch for one read (align.cpp:591-606); readset 0 = SE, 1/2 = mates */
int bso_se_align(bso_aligner *a, uint32_t index, int readset, const char *seq, const char *qual,
                 bso_read_result *out);
const bso_hit *bso_se_hits(const bso_aligner *a, int orient, int w);
/* body of PairAlign::Do_Batch for one pair (pairs.cpp:192-218) */
int bso_pe_align(bso_aligner *a, bso_aligner *b, uint32_t index, const char *seq_a, const char *qual_a,
                 const char *seq_b, const char *qual_b, bso_pair_result *out);
const bso_pair *bso_pe_pairs(const bso_aligner *a, int w);
/* work counters: header lookups, candidates, 64-bit reference words touched */
void bso_counters(const bso_aligner *a, uint64_t *n_lookup, uint64_t *n_cand, uint64_t *sum_w, uint64_t *n_orient);
uint32_t bso_myrand(const bso_params *p, uint32_t index, uint32_t *rseed); /* utilities.cpp:40-50 */

/* batch helpers used by bench.py's cpu_baseline leg and the tests: SoA reads, n_threads pthreads
 * pulling fixed-size chunks like main.cpp:49-73.  seqs is a flat buffer, read i = seqs[off[i]..off[i+1]).
 * quals may be NULL.  results[i] filled in input order.  Returns 0. */
int bso_se_batch(const bso_params *p, const bso_ref *r, uint32_t n_reads, const char *seqs,
                 const uint64_t *off, const char *quals, uint32_t first_index, int n_threads,
                 bso_read_result *results, uint64_t counters[4]);
int bso_pe_batch(const bso_params *p, const bso_ref *r, uint32_t n_pairs, const char *seqs_a,
                 const uint64_t *off_a, const char *quals_a, const char *seqs_b, const uint64_t *off_b,
                 const char *quals_b, uint32_t first_index, int n_threads, bso_pair_result *results,
                 uint64_t counters[4]);
/* the same with the aligner objects in leak_mode (1: planner state runs through the input in order, as with `bsmap -p 1`;
 * the chunks re-establish it by a planner-only replay, see batch_job in bsx_oracle.c) */
int bso_se_batch_leak(const bso_params *p, const bso_ref *r, uint32_t n_reads, const char *seqs,
                      const uint64_t *off, const char *quals, uint32_t first_index, int n_threads, int leak_mode,
                      bso_read_result *results, uint64_t counters[4]);
int bso_pe_batch_leak(const bso_params *p, const bso_ref *r, uint32_t n_pairs, const char *seqs_a,
                      const uint64_t *off_a, const char *quals_a, const char *seqs_b, const uint64_t *off_b,
                      const char *quals_b, uint32_t first_index, int n_threads, int leak_mode,
                      bso_pair_result *results, uint64_t counters[4]);
#ifdef __cplusplus
}
#endif
#endif
