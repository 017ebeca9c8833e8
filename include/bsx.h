/* bsx.h — C ABI of libbsx.so, the MI355X (gfx950) implementation of the BSMAP v2.6 alignment hot path.
 *
 * BSMAP has no plugin/FFI interface; the seam this library replaces is the per-thread C++ object call
 *     a.ImportBatchReads(n, reads); a.Do_Batch(ref);          (reference main.cpp:57-64, :94-102)
 * plus the two start-up calls that produce the data Do_Batch reads:
 *     ref.Run_ConvertBinseq(fin_db); ref.CreateIndex();        (reference main.cpp:462, :174-178)
 * Every entry point below cites the reference interface it stands in for (file:line in the BSMAP
 * v2.6 tree).  Signatures carry plain pointers and sizes only; all device memory is owned by the
 * library behind opaque handles; no call throws or aborts — errors are negative return codes.
 *
 * Host text formatting (SAM/BSP, reference align.cpp:631-765, pairs.cpp:288-498) stays on the host
 * side of this boundary and consumes the numeric records returned here.
 */
#ifndef BSX_H
#define BSX_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BSX_MAXSNPS 15    /* reference param.h:27 */
#define BSX_MAXHITS 1000  /* reference makefile:4 (-DMAXHITS) */
#define BSX_MAX_READLEN 144 /* reference param.cpp:80 (READ_144) */

enum {
    BSX_OK = 0,
    BSX_ERR_ARG = -1,      /* bad argument / option value (reference: cerr + exit(1), main.cpp:260-265) */
    BSX_ERR_IO = -2,       /* cannot open / parse file (reference main.cpp:457-461) */
    BSX_ERR_NOMEM = -3,    /* host or device allocation failed */
    BSX_ERR_DEVICE = -4,   /* HIP runtime error (see bsx_last_error_detail) */
    BSX_ERR_STATE = -5,    /* call order violated (e.g. align before index build) */
    BSX_ERR_LIMIT = -6,    /* reference limits exceeded (>= 2^32 nt, -v > 15, -w > 1000, -s > 16) */
    BSX_ERR_NODEVICE = -7  /* no usable gfx950 device: the library has NO CPU fallback */
};
const char *bsx_strerror(int code);
const char *bsx_last_error_detail(void); /* thread-local text of the last failing HIP call */

/* ---- parameters: reference class Param (param.h:54-169), option parser main.cpp:234-289 ------- */
typedef struct bsx_params {
    int32_t seed_size;          /* -s  */
    int32_t index_interval;     /* -I  */
    int32_t max_snp_num;        /* -v  */
    int32_t max_num_hits;       /* -w  */
    int32_t chains;             /* -n  */
    int32_t pairend;            /* -b given */
    int32_t min_insert;         /* -m  */
    int32_t max_insert;         /* -x  */
    int32_t report_repeat_hits; /* -r  */
    int32_t randseed;           /* -S  (0 = nondeterministic in the reference; here: treated as seed 0 of the same hash) */
    int32_t qual_threshold;     /* -q  */
    int32_t zero_qual;          /* -z  */
    int32_t max_ns;             /* -f  */
    int32_t max_readlen;        /* -L  */
    int32_t out_sam;            /* affects TrimLowQual's quality rebasing only (align.cpp:64-67) */
    int32_t rrbs;               /* set by bsx_params_set_digest */
    int32_t digest_pos;
    int32_t n_adapter;          /* -A (up to 10) */
    char digest_site[32];
    char adapter[10][128];
    char read_nt, ref_nt;       /* -M */
    char pad_[2];
    /* derived (bsx_params_finish) */
    uint8_t bit_nt[4];          /* 2-bit code of A,C,G,T (Param::SetAlign, param.cpp:187-231) */
    uint8_t profile_a[BSX_MAXSNPS + 1][16]; /* SeedProfile.a (param.cpp:85-93) */
    uint32_t seed_bits;
    int32_t max_seedseg_num;
    uint32_t total_kmers;
} bsx_params;

int bsx_params_default(bsx_params *p);                         /* Param::Param(), param.cpp:6-83 */
int bsx_params_set_digest(bsx_params *p, const char *site);    /* Param::SetDigestionSite, param.cpp:95-106 ("C-CGG") */
int bsx_params_finish(bsx_params *p);                          /* SetAlign + SetSeedSize + InitMapping; validates limits */

/* ---- reference genome: RefSeq::Run_ConvertBinseq (dbseq.cpp:215-282) --------------------------- */
typedef struct bsx_ref bsx_ref;
int bsx_device_count(void);
/* NUMA node the device hangs on (from its PCI address; -1 if unknown): a host program under a CPU quota does well to keep its threads
 * and buffers on that node (bsmap_amd/csrc/bsx_cpus.h) */
int bsx_device_numa_node(int device);
/* parse FASTA text on the host with the reference's tokenisation, pack both strands, upload to `device` */
int bsx_ref_create_from_fasta(const bsx_params *p, const char *text, uint64_t n_bytes, int device, bsx_ref **out);
int bsx_ref_create_from_file(const bsx_params *p, const char *path, int device, bsx_ref **out);
/* deterministic synthetic genome generated ON the device (bench workload) */
int bsx_ref_create_synthetic(const bsx_params *p, uint32_t n_chr, const uint32_t *chr_len, uint64_t seed, int device, bsx_ref **out);
/* text of a synthetic chromosome, positions [from, from+n) (lets tests run the oracle's packer on the same sequence) */
int bsx_synth_chr_text(const bsx_ref *r, uint32_t c, uint32_t from, uint32_t n, char *out);
int bsx_ref_packed_on_device(const bsx_ref *r);   /* 1: the FASTA text was uploaded and packed by kernels (line-regular text of a WGBS reference, csrc/bsx_pack.hip); 0: by the host packer.  Same words either way */
void bsx_ref_destroy(bsx_ref *r);
uint32_t bsx_ref_n_chr(const bsx_ref *r);
uint64_t bsx_ref_n_words(const bsx_ref *r);            /* words per strand copy incl. 2*400 margin (dbseq.h:15) */
uint32_t bsx_ref_n_blocks(const bsx_ref *r);
/* copies out: anchor[n_chr+1], chr_size[n_chr], rc_offset[n_chr]  (RefSeq::ref_anchor, RefTitle, dbseq.h:24-30,113) */
int bsx_ref_info(const bsx_ref *r, uint32_t *anchor, uint32_t *chr_size, uint32_t *rc_offset);
const char *bsx_ref_chr_name(const bsx_ref *r, uint32_t c);
int bsx_ref_blocks(const bsx_ref *r, uint32_t *id, uint32_t *begin, uint32_t *end); /* RefSeq::_blocks, sorted */
int bsx_ref_download_words(const bsx_ref *r, uint32_t *refcat, uint32_t *crefcat);  /* device -> host, n_words each */

/* ---- seed index: RefSeq::CreateIndex (dbseq.cpp:516-539) ---------------------------------------- */
/* (WGBS, -I <= 4: the build can also store, per entry, the 32 reference nt left and right of the entry's seed — the CONTEXT TABLE, 16 bytes per entry,
 *  23.6 GB at hg38 size — for the main kernel's context prefilter, which runs where the work counters are off.  It buys speed, never results, and it must
 *  not take the memory the batches need:
 *    bsx_ref_set_context(r, mode, headroom)  before bsx_index_build.  mode 0: never; 1 (default): only if `headroom` bytes of device memory stay free
 *        behind it (0 = the default, 32 GiB: the fixed part of a 2^22-pair batch — 20.6 GB of per-wave slabs and per-unit arrays — plus the pools'
 *        reserve); 2: whenever it can be allocated, and never dropped.
 *    bsx_ref_context_bytes(r)                what the built index holds (0: none).
 *    bsx_ref_drop_context(r)                 frees it (BSX_ERR_STATE while a batch of the reference exists: a running kernel may hold its address).
 *  In modes 0 / 1 bsx_batch_create itself drops the table and tries once more when the FIRST batch of a reference does not fit beside it.
 *  Test hook: BSX_CTX=0 in the environment builds none.) */
int bsx_ref_set_context(bsx_ref *r, int mode, uint64_t headroom_bytes);
uint64_t bsx_ref_context_bytes(const bsx_ref *r);
int bsx_ref_drop_context(bsx_ref *r);
int bsx_index_build(bsx_ref *r);                        /* built on the GPU; entry order identical to the reference */
uint64_t bsx_index_n_entries(const bsx_ref *r);
/* CSR copy-out: bucket_off[total_kmers+1], bucket_nfwd[total_kmers], entries[n_entries]
 * (WGBS: u32 global positions, RefSeq::index2; RRBS: pairs {tag,loc}, RefSeq::index — 2 words per entry) */
int bsx_index_download(const bsx_ref *r, uint32_t *bucket_off, uint32_t *bucket_nfwd, uint32_t *entries);
uint32_t bsx_ref_n_sites(const bsx_ref *r, uint32_t c);                /* RRBS: RefSeq::CCGG_sites */
int bsx_ref_sites(const bsx_ref *r, uint32_t c, uint32_t *sites);

/* ---- read batches: SingleAlign / PairAlign ImportBatchReads + Do_Batch -------------------------- */
typedef struct bsx_batch bsx_batch;

/* per-read record == what SingleAlign::StringAlign (align.cpp:610-627) hands to the formatter */
typedef struct bsx_hit {
    uint32_t chr;        /* 2*c for the + reference strand copy, 2*c+1 for the - copy (Hit.chr) */
    uint32_t loc;        /* 0-based forward coordinate of the first base (Hit.loc) */
    uint16_t n_best;     /* hits in the best class, both read orientations (<= -w) */
    int8_t best_class;   /* mismatches of the best class; -1 = no hit */
    uint8_t flags;       /* BSX_F_* */
    uint8_t len;         /* read length after trimming (FilterReads) */
    uint8_t max_snp;     /* read_max_snp_num (align.cpp:586) */
    uint8_t seedseg;     /* seedseg_num (align.cpp:440) */
    uint8_t raw_len;     /* length before trimming */
} bsx_hit;
#define BSX_F_FILTERED 1u  /* FilterReads() rejected the read (QC) */
#define BSX_F_CHAIN 2u     /* chosen hit is on the read's reverse-complement orientation (chits) */
#define BSX_F_LIMIT 4u     /* single-end RRBS only: the read matched more than 2^18 distinct places inside the threshold; the set that
                              suppresses duplicate hits overflowed and the record may differ from the reference's */

/* optional per-read class counts: _cur_n_hit[w] / _cur_n_chit[w] (align.h:85-86) */
typedef struct bsx_class_counts { uint16_t n_hit[BSX_MAXSNPS + 1], n_chit[BSX_MAXSNPS + 1]; } bsx_class_counts;

/* per-pair record == PairHit (pairs.h:13-20) chosen by StringAlignPair (pairs.cpp:222-242), plus the
 * two single-mate records StringAlignUnpair (pairs.cpp:244-286) would use */
typedef struct bsx_pair {
    uint32_t a_chr, a_loc, b_chr, b_loc;
    int32_t insert;
    uint16_t n_pairs;    /* pairs in the best pair class */
    int8_t pair_class;   /* na+nb of the best class, -1 none */
    uint8_t chain;       /* PairHit.chain */
    uint8_t na, nb;
    uint8_t paired;      /* PairAlign::RunAlign return value (level+1, 0 = none) */
    uint8_t unpaired_out;/* 1: pair not reported (tmp==1 || paired==0): use a/b below */
    uint32_t pad_;
    bsx_hit a, b;        /* per-mate records; n_best is ma/mb of StringAlignUnpair when unpaired_out */
} bsx_pair;

/* work counters, SURVEY §8(d): 0 n_lookup, 1 n_cand, 2 sum_w (64-bit reference words the reference
 * algorithm touches), 3 n_orient, 4 reads/pairs processed, 5 aligned reads (n_aligned semantics), 6 aligned pairs,
 * 7 candidates evaluated by the scan kernel of the heavy pipeline (k_hscan; a subset of 1 plus the little it evaluates
 * speculatively), 8 their reference words (as 2), 9 / 10 how many of them stopped after the first word / went through all
 * five, 11-14 the share of 0-3 that the main kernel (k_align) did itself — units it did not hand to the heavy pipeline —, so
 * that each kernel's algorithmic bytes can be recomputed from the counters, 15 the part of 7 that was evaluated in groups of two and more
 * tasks over one window and read offset (k_hscan_same: one fetch and shift of the candidates' reference for up to 16 reads), 16 records flagged BSX_F_LIMIT
 * (single-end RRBS reads that matched more than 2^18 distinct places: the one capacity the reference does not have — 0 on every workload seen) */
#define BSX_N_COUNTERS 17

int bsx_batch_create(bsx_ref *r, uint32_t max_units, int paired, bsx_batch **out);
void bsx_batch_destroy(bsx_batch *b);
/* Page-locked host buffers for the arrays handed to bsx_batch_upload_* / bsx_batch_results_*: with them the transfers
 * are plain DMA (the reference has no counterpart: its reads never leave host memory).  Ordinary malloc memory works too. */
int bsx_thread_device(int device);   /* make `device` the calling thread's current GPU (for bsx_pinned_alloc from a helper thread) */
void *bsx_pinned_alloc(size_t bytes);
void bsx_pinned_free(void *p);

/* SoA upload (ImportBatchReads, align.cpp:42-46 / pairs.cpp:27-32): read i = seqs[off[i] .. off[i+1]);
 * quals may be NULL (FASTA input: constant default quality, reads.cpp:108); reads longer than -L are truncated
 * (reads.cpp:115-117).  first_index = ReadInf.index of unit 0 (reads.cpp:97). */
int bsx_batch_upload_se(bsx_batch *b, uint32_t n, const char *seqs, const uint64_t *off, const char *quals, uint32_t first_index);
int bsx_batch_upload_pe(bsx_batch *b, uint32_t n, const char *seqs_a, const uint64_t *off_a, const char *quals_a,
                        const char *seqs_b, const uint64_t *off_b, const char *quals_b, uint32_t first_index);
/* synthetic bisulfite reads sampled ON the device from the resident reference (bench workload) */
int bsx_batch_synth_reads(bsx_batch *b, uint32_t n, uint32_t read_len, uint64_t seed, uint32_t first_index);
/* the same for the other bench workloads: kind 0 plain; 1 trimming workload — qualities with a 3' tail of 10-60 nt
 * at Q2-Q15, 30 % of the fragments 30-149 nt long so that the reads run into the adapter AGATCGGAAGAGC...; 2 RRBS — single
 * reads that start at the digestion sites of an RRBS reference, fragments read_len..220 nt */
int bsx_batch_synth_reads_kind(bsx_batch *b, uint32_t n, uint32_t read_len, uint64_t seed, uint32_t first_index, int kind);
/* Do_Batch (align.cpp:591-606 / pairs.cpp:192-218).  The main kernel is queued on the batch's HIP stream.  Units it hands to the heavy
 * pipeline (reads whose candidate lists run into the tens of thousands) are then taken through passes of control and scan kernels that
 * are chained on the device; the calling thread queues those passes and looks at one counter per two passes to learn when the batch
 * is finished — it SLEEPS on an event meanwhile, it does not spin — so the call returns when the last pass is queued, which for such a
 * batch is close to the end of its device work.  bsx_batch_sync waits for the rest.  To keep a GPU busy through the latency-bound tail
 * of a batch, run two batches per GPU from two host threads (bench.py, the command line): their kernels interleave on the device. */
int bsx_batch_run(bsx_batch *b);
/* the same over units [first_unit, first_unit+n_units) of the uploaded batch (ReadInf.index = first_index + unit) */
int bsx_batch_run_range(bsx_batch *b, uint32_t first_unit, uint32_t n_units);
int bsx_batch_sync(bsx_batch *b);
/* "-p 1 exact" mode (off by default).  The reference never resets SingleAlign::seed_start_offset / seed_array (align.h:82-91):
 * a read with (len - I + 1) % S == 0 is planned with what earlier reads of its stream left there, so with -p 1 its result is a
 * deterministic function of the reads before it (with -p > 1 it depends on thread scheduling).  By default every read starts
 * from zeroed planner state; with this mode on, such reads look their predecessors up — units of the batch,
 * then the history below — and reproduce the single-threaded reference exactly.  History = the reads that precede unit 0 in
 * the input (the tail of the previous batch, up to 65536 reads; quals NULL iff the batch has none). */
int bsx_batch_set_leak_exact(bsx_batch *b, int on);
int bsx_batch_set_history(bsx_batch *b, uint32_t n, const char *seqs_a, const uint64_t *off_a, const char *quals_a, const char *seqs_b,
                          const uint64_t *off_b, const char *quals_b);
/* The same state as an explicit value, for callers that cut an input into batches: `bsx_batch_get_leak_state` returns the planner state
 * behind the last read of the batch's streams (history, then every unit of the batch; a pure function of those reads and of the state
 * before them), `bsx_batch_set_leak_state` makes a state the one before the streams' first read (null: the zero state of a fresh
 * SingleAlign object).  Chaining get -> set from batch to batch reproduces `bsmap -p 1` for any input, however far back the read
 * that set a value lies; no history needs to be attached then.  BSX_LEAK_STATE_BYTES bytes, opaque. */
#define BSX_LEAK_STATE_BYTES 2576
int bsx_batch_set_leak_state(bsx_batch *b, const void *state, size_t bytes);
int bsx_batch_get_leak_state(bsx_batch *b, void *state, size_t bytes);
float bsx_batch_kernel_ms(bsx_batch *b);          /* HIP-event time of the last run's align kernel (after sync) */
/* the scan kernel's launches of the last run: their number and the sum of their HIP-event durations on the stream they
 * were launched on (control kernels of the other unit group may run beside them on another stream).  Passes are queued two
 * chunks ahead of what the host knows (bsx_batch_run), so the count includes the launches queued behind a group's last pass:
 * they find no task, their grids exit at once (tens of microseconds each) and they are part of the sum. */
int bsx_batch_scan_ms(bsx_batch *b, float *total_ms, uint32_t *launches);
/* HIP-event times of the last run by stage, in ms (after sync): out4 = {main kernel k_align (with the exact mode's pre-pass), control kernel k_hctrl summed over
 * its passes, the order kernels between a control pass and its scan, the scan kernel summed over its launches}; each on the stream the kernel is launched on.
 * With ONE batch in flight and one unit group (bench.py's serial replay) no two of them overlap and they add up to the run; with several batches in flight the
 * durations include what ran beside them. */
int bsx_batch_set_stage_timing(bsx_batch *b, int on);   /* off by default: the per-pass events are only recorded for runs queued while it is on */
int bsx_batch_stage_ms(bsx_batch *b, float out4[4], uint32_t *control_passes /* may be NULL */);
int bsx_batch_results_se(bsx_batch *b, bsx_hit *out, bsx_class_counts *counts /* may be NULL */);
int bsx_batch_results_pe(bsx_batch *b, bsx_pair *out, bsx_class_counts *counts_a, bsx_class_counts *counts_b, uint16_t *n_pairs31);
int bsx_batch_counters(bsx_batch *b, uint64_t c[BSX_N_COUNTERS]);   /* accumulated since creation / last reset */
/* Work counters (on by default).  Counters 1-2 and 7-10 count the work the REFERENCE would do on the same reads — candidates, and the
 * 64-bit reference words CountMismatch would touch given its two early-outs (align.h:189-197); SURVEY §8(d)'s algorithmic-byte formula is
 * evaluated with them and the parity suite compares them with the oracle's.  Classifying every candidate of the heavy pipeline's scan by
 * those early-outs is a quarter of the scan kernel's vector instructions and changes no hit: with the counters off the scan kernels skip
 * it (and the control kernel its count-only walks) — every record is identical, counters 0-3 and 7-15 are then incomplete, 4-6 (aligned
 * reads / pairs) stay exact.  The command line runs with them off; bench.py times both and says which is which. */
int bsx_batch_set_work_counters(bsx_batch *b, int on);
int bsx_batch_reset_counters(bsx_batch *b);
/* download the device-resident input reads (for the CPU baseline on device-synthesised input) */
int bsx_batch_download_reads(bsx_batch *b, int mate, char *seqs, uint64_t *off);
int bsx_batch_download_quals(bsx_batch *b, int mate, char *quals);   /* device-synthesised qualities (kind 1), same offsets */
/* test hooks: mode 1 keeps every hit / pair list of every unit (needs max_units small; layout in DESIGN.md);
 * mode 2 records the shader-clock cycles each unit took (diagnostic runs only); 0 switches both off */
int bsx_batch_set_debug(bsx_batch *b, int mode);
int bsx_batch_unit_cycles(bsx_batch *b, uint32_t *cycles_per_unit);
int bsx_batch_ctrl_clocks(bsx_batch *b, uint64_t clocks[24]); /* mode 2: heavy control kernel, wave cycles by category: [0..7] sums, [8..15] longest single span, [16..21] break-down of the longest visit (cycles << 16 | spans) */
int bsx_batch_debug_hits(bsx_batch *b, uint32_t unit, int mate, int orient, int w, uint32_t *chr_loc_pairs, uint32_t cap);
int bsx_batch_debug_pairs(bsx_batch *b, uint32_t unit, int w, uint32_t *pairhits6 /* chain|na<<16|nb<<24, insert, a.chr, a.loc, b.chr, b.loc */, uint32_t cap);
int bsx_batch_debug_plan(bsx_batch *b, uint32_t unit, int mate, int32_t *start_arrays32 /* [2][16] */, int32_t *seedindex32 /* [2][16] */);
/* tuning knob: resident waves per CU for the persistent align kernel (default chosen from register use) */
int bsx_set_waves_per_cu(int waves);
/* tuning knob: candidate-list length (one SnpAlign call, one read orientation) from which a unit is handed to the
 * heavy pipeline (chip-wide scan tasks + resumable control passes); 0 = the library's choice by mode (32768 WGBS, 4096 RRBS),
 * a value no list reaches = never.  Results do not depend on it. */
int bsx_set_heavy_threshold(int n_candidates);
/* pool sizes of the heavy pipeline for batches created afterwards (the defaults follow the batch: bsx_default_heavy_limits returns them — up to 262144 units per round
 * (64 GB of slabs) and 2^22 scan tasks for a 2^22-pair batch —, and bsx_batch_pool_sizes what a batch ended up with); small values only make it take more rounds / passes —
 * used by the tests to exercise those paths; (0, 0) restores the defaults.  The halving below never raises a value set here. */
int bsx_set_heavy_limits(uint32_t units_per_round, uint32_t task_pool);
/* Either way those are STARTING sizes: a batch halves its pools until they fit the device's free memory with a reserve to spare —
 * by default room for one more batch like itself (per-wave slabs + per-unit arrays) plus 4 GB; bsx_set_pool_reserve(bytes) sets the
 * reserve for batches created afterwards (0 = that default).  bsx_batch_pool_sizes returns what a batch ended up with. */
int bsx_set_pool_reserve(uint64_t bytes);
/* the library's own starting sizes for runs of `units` units (a caller whose batches HOLD more units than a run covers — bench.py keeps a ring of
 * steps resident and runs one step at a time with bsx_batch_run_range — passes these to bsx_set_heavy_limits: a batch sizes its pools for the
 * units it was created for) */
int bsx_default_heavy_limits(const bsx_params *p, uint32_t units, int paired, uint32_t *units_per_round, uint32_t *task_pool);
int bsx_batch_pool_sizes(const bsx_batch *b, uint32_t *units_per_round, uint32_t *task_pool);
/* Device bytes a batch of `max_units` will allocate, computed on the host (no device needed): out3 = {per-unit arrays (reads, offsets,
 * records), the main kernel's per-wave slabs for a grid of n_cu x blocks_per_cu blocks, work pools at their starting size}.
 * bench.py checks its whole plan against the device's memory with it before it allocates anything. */
int bsx_batch_plan_bytes(const bsx_params *p, uint32_t max_units, int paired, uint64_t n_entries, uint32_t n_cu, uint32_t blocks_per_cu, uint64_t *out3);
int bsx_batch_last_heavy_units(bsx_batch *b);   /* units the last run handed to the heavy pipeline */
int bsx_batch_last_heavy_list(bsx_batch *b, uint32_t *units, uint32_t cap);   /* their unit numbers inside the batch, in the order the main kernel deferred them (at most cap; returns how many): the tests use it to aim the oracle at them */
int bsx_batch_last_redo_units(bsx_batch *b);    /* of those, units the main kernel had to redo (their duplicate set outgrew the heavy slab; single-end RRBS) */

/* ---- measurement aid ("report both peak and a measured ceiling") -----------------------------------
 * Memory rates of `device` in GB/s (1e9 bytes): streaming read of `bytes`, streaming copy (read + write counted), and
 * 16-byte loads at random 4-byte-aligned addresses inside a window of `gather_window_bytes` — the access pattern of the
 * candidate scan (useful bytes = 16 per load; *_Gloads_per_s = 1e9 loads/s).  Not on the alignment path. */
int bsx_probe_memory(int device, uint64_t bytes, uint64_t gather_window_bytes, double *read_GBps, double *copy_GBps, double *gather16_GBps,
                     double *gather16_Gloads_per_s);

/* ---- methylation-ratio pile-up (reference: methratio.py of the BSMAP tree) ---------------------------
 * The reference walks the alignments in Python and increments two per-position counters; here that part runs on the
 * GPU with atomic counters in HBM.  The caller (bsmap_amd/methratio.py) keeps the reference's option parsing, FASTA and
 * BSP/SAM line parsing (get_alignment's filters, methratio.py:31-48) and prints the table (methratio.py:130-151). */
typedef struct bsx_meth bsx_meth;
/* meth/depth arrays for chromosomes of the given lengths (methratio.py:79-83); rm_dup != 0 also allocates the
 * fragment-end table of -r (methratio.py:50-54) */
int bsx_meth_create(uint32_t n_chr, const uint64_t *chr_len, int rm_dup, int device, bsx_meth **out);
/* the same from the reference FASTA (methratio.py:67-77: name = first token of the header, lines stripped and joined, upper case);
 * chroms_csv = the -c list or NULL.  The names are kept: chr_names / order may then be NULL in the calls below. */
int bsx_meth_create_from_fasta(const char *path, const char *chroms_csv, int rm_dup, int device, bsx_meth **out);
uint32_t bsx_meth_n_chr(const bsx_meth *m);
const char *bsx_meth_chr_name(const bsx_meth *m, uint32_t chr);
void bsx_meth_destroy(bsx_meth *m);
int bsx_meth_set_reference(bsx_meth *m, uint32_t chr, const char *upper_seq /* chr_len[chr] letters, upper case */);
/* n alignments in input order: chromosome id, 0-based position, strand code (0 "++", 1 "-+", 2 "+-", 3 "--"), insert size
 * (BSP column 8 / SAM TLEN), cut_at (SAM with insert > 0: PNEXT-1, methratio.py:64; otherwise -1), read letters as
 * concatenated bytes + offsets.  Duplicate removal (first alignment in input order wins), fill-in trimming with
 * trim_fillin, the bounds test and the counter updates of methratio.py:50-63,100-114 happen on the device. */
int bsx_meth_add(bsx_meth *m, uint32_t n, const uint32_t *chr, const int64_t *pos, const uint8_t *strand, const int32_t *insert, const int64_t *cut_at,
                 const char *seqs, const uint64_t *seq_off, uint32_t trim_fillin);
/* the same for a whole BSMAP mapping file (sam = 0 BSP text, 1 SAM text, 2 BAM): the file is memory-mapped and parsed by host
 * threads with get_alignment's filters (NM/QC or unmapped, -u unique, -p pair, chromosome known; methratio.py:31-48), alignments
 * keep the file's order; chr_names[n_chr] in id order */
int bsx_meth_add_file(bsx_meth *m, const char *path, int sam, const char *const *chr_names, int unique, int pair, uint32_t trim_fillin, uint64_t *n_lines);
int bsx_meth_combine_cpg(bsx_meth *m);                      /* -g, methratio.py:118-128 */
int bsx_meth_valid_mappings(bsx_meth *m, uint64_t *n);      /* "total %d valid mappings" */
/* rows of one chromosome's table in position order: positions with depth >= min_depth and (methylated > 0 or meth0);
 * n_covered / sum_depth count every position with depth >= min_depth (methratio.py:138-142) */
int bsx_meth_report_chr(bsx_meth *m, uint32_t chr, uint32_t min_depth, int meth0, uint32_t *n_rows, uint64_t *n_covered, uint64_t *sum_depth);
int bsx_meth_fetch_rows(bsx_meth *m, uint32_t *pos0, uint32_t *depth, uint32_t *meth);   /* the rows of the last report_chr */
/* the whole table file of methratio.py:130-151 (header, chromosomes in the given order — the reference sorts the names —,
 * ratio and Wilson interval with the reference's arithmetic, "%.3f"); returns the counts of the summary line */
int bsx_meth_write_table(bsx_meth *m, const char *path, uint32_t n_order, const uint32_t *order, const char *const *chr_names, uint32_t min_depth, int meth0,
                         uint64_t *n_covered, uint64_t *sum_depth);

#ifdef __cplusplus
}
#endif
#endif
