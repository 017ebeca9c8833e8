// bsx_kernel_args.h — argument block of the align kernel (one struct passed by value).
#pragma once
#include "bsx_internal.h"

struct AlignArgs {
    DevParams P;
    uint32_t n_units, first_index;   // units [first_unit, n_units) are processed; ReadInf.index = first_index + unit
    uint32_t first_unit;
    int32_t debug;             // 1: scratch slab per unit (lists kept for inspection) instead of per wave
    uint32_t rowcap;           // capacity of one hit / pair list row (= -w + 64, see DESIGN.md "cap overshoot")
    uint32_t kcap, hbits;      // per-mate duplicate-suppression set of a slab: key capacity, log2 of its hash slots (Slab in bsx_align.hip)
    uint32_t hkcap, hhbits;    // the same for the slabs of deferred units (heavy pipeline)
    uint64_t hslab_bytes;
    uint32_t *redo_list, *redo_count;  // deferred units whose small set overflowed: redone by the main kernel (unit_list)
    const uint32_t *unit_list; // main kernel: process units unit_list[first_unit .. n_units) instead of the range itself
    const uint8_t *seq[2];     // ASCII reads, mate 0 / 1
    const uint64_t *off[2];    // [n_units+1] byte offsets
    const uint8_t *qual[2];    // may be null
    bsx_hit *hits_out;
    bsx_pair *pairs_out;
    bsx_class_counts *cc[2];   // may be null
    uint16_t *npairs_out;      // [n_units][32], may be null
    uint8_t *scratch;
    uint64_t slab_bytes;
    uint32_t work_counters;    // 1: the scan kernels also classify every candidate by the reference's two early-outs (align.h:189-197) so that counters 1-2 and 7-10
                               //    equal the work the reference would do (the parity suite compares them with the oracle's); 0: hits only — n_cand / sum_w are
                               //    then incomplete (bsx_batch_set_work_counters)
    uint32_t heavy_threshold;  // candidate-list length from which a unit is deferred to the heavy pipeline (0 = never)
    uint32_t *heavy_list;      // [n_units] unit ids deferred by the main kernel
    uint32_t *heavy_count;
    uint32_t *queue;           // work queue head of the main kernel (zeroed before launch)
    uint64_t *counters;        // BSX_N_COUNTERS
    uint64_t *scan_stats;      // [64][8] sharded statistics of the scan kernel: candidates, reference words, class-one, class-five (counters 7-10)
    uint32_t *dbg_cycles;      // [n_units] shader-clock cycles spent on each unit (diagnostic builds of a run only), may be null
    uint64_t *dbg_cat;         // [16] category clocks (sums, then the longest single span of each) of the heavy control kernel (diagnostic runs only), may be null
    uint8_t *dbg_plan;         // [n_units][128]: start[2][16], order[2][16] for mate a then mate b
    // "-p 1 exact" mode (bsx_batch_set_leak_exact): reads whose planner state the reference inherits from earlier reads of the
    // same stream (align.h:82-91, never reset) look those reads up — in this batch, then in the history the caller attached
    int32_t leak_exact;
    const uint8_t *leak_rec;   // [n_units][2] LeakRec written by k_leak_resolve (exact mode only)
    uint32_t n_hist;           // reads of history per mate stream (the reads that precede unit 0 in the input)
    uint32_t n_units_all;      // units of the whole batch (a run may cover a range; the stream of the exact mode is the whole batch)
    uint16_t *leak_meta[2];    // per stream position (history, then units): trimmed length, 0xffff = rejected by FilterReads
    uint32_t *leak_blkmax[2], *leak_blkset[2];  // per block of 4096 positions: most seed offsets of an unfiltered read / a read sets the offset
    const void *leak_init;     // LeakState before the stream's first read (null: zero)
    const uint8_t *hist_seq[2];
    const uint64_t *hist_off[2];
    const uint8_t *hist_qual[2];
};

// The three counters of a control pass — active units it leaves, scan tasks it publishes, its work queue — take an atomic per visit each.  One memory word (one
// L2 channel) serves about 88 atomics per microsecond: side by side in one cache line, as they were through round 5, the 3 x 179 K atomics of an RRBS pass were
// 6 ms of its 6.4 (C4 spent as long in k_hctrl as in its scans).  A counter block keeps them 4 KB + 256 B apart (different channels), and k_hctrl takes queue
// entries in chunks and hands in its active units in batches (bsx_align.hip).
#ifndef BSX_HCNT_TASKS   /* (tools/build_variant.sh: -DBSX_HCNT_TASKS=1 -DBSX_HCNT_QUEUE=2 -DBSX_HCTRL_BATCH=0 is the round-5 arrangement) */
#define BSX_HCNT_TASKS 1088u   /* word offset of the task counter inside a counter block */
#define BSX_HCNT_QUEUE 2176u   /* ... of the queue head */
#endif
#define BSX_HCNT_BLOCK 3072u   /* words per counter block */
#ifndef BSX_HCTRL_BATCH
#define BSX_HCTRL_BATCH 1      /* k_hctrl: queue entries in chunks, active units handed in 32 at a time */
#endif

// heavy pipeline (see bsx_align.hip): untyped view used by the host side
struct HeavyArgsRaw {
    uint8_t *state;            // [cap] HState
    uint8_t *slabs;            // [cap] one scratch slab per deferred unit of the current round
    uint32_t *active_in, *active_out, *n_active_out;
    uint8_t *tasks, *tout;     // [task_cap] HTask / HTaskOut
    uint32_t *n_tasks;
    uint32_t *queue;           // [2] work queue heads of k_hctrl and k_hscan
    const uint32_t *n_active_in_ptr;  // device count of active_in (passes after the first)
    uint32_t n_active_in, task_cap, fresh, list_base, hidx_base;
    const uint32_t *order;     // task ids in scan order (sorted by the index entry they start at), or null
    uint32_t xcd_map;
    const uint32_t *ghead;           // k_hscan_same: group sizes by scan slot (bsx_launch_task_order with groups writes them into the rank array)
    uint32_t *glist;                 // k_hscan_same: start slots of the groups [task_cap], then their two counts
};

void bsx_launch_align(const AlignArgs &A, int paired, int grid_blocks, hipStream_t stream);
void bsx_launch_leak(const AlignArgs &A, int paired, int n_cu, bool with_meta, bool resolve, void *final_out, hipStream_t stream);
size_t bsx_leakrec_bytes(void);
size_t bsx_leakstate_bytes(void);
uint32_t bsx_leak_blk(void);
void bsx_launch_hctrl(const AlignArgs &A, const HeavyArgsRaw &H, int paired, int grid_blocks, hipStream_t stream);
void bsx_launch_hscan(const AlignArgs &A, const HeavyArgsRaw &H, hipStream_t stream, uint32_t max_tasks = 0);         // grid sized for H.task_cap (or max_tasks: the blocks sweep); the count stays on the device
void bsx_launch_hscan_same(const AlignArgs &A, const HeavyArgsRaw &H, hipStream_t stream, uint32_t max_tasks = 0);    // WGBS: tasks of one window and read offset share fetch and shift
void bsx_launch_hscan_shared(const AlignArgs &A, const HeavyArgsRaw &H, hipStream_t stream, uint32_t max_tasks = 0);  // RRBS: up to 16 tasks of one window per wave
// scan order of a pass (task ids by the index entry they start at, 2^shift entries per bin), computed on the device
uint32_t bsx_bin_chunks(uint32_t n_bins);
void bsx_launch_task_order(const HeavyArgsRaw &H, uint32_t shift, uint32_t n_bins, uint32_t *bins, uint32_t *bstart, uint32_t *chunk_tot, uint32_t *rank, uint32_t *order,
                           uint32_t *zero_blk, hipStream_t stream, uint32_t spread = 0, bool groups = false);   // spread: tasks inside one sub-range are dealt over the bins they cover by their read offset (k_hscan_same)
void bsx_sig_hist_pass(const HeavyArgsRaw &H, hipStream_t stream);   // diagnostics, BSX_SIGHIST=1
void bsx_sig_hist_report(void);
void bsx_sector_pass(const bsx_ref *r, hipStream_t stream);   // diagnostics (build with -DBSX_SECTOR_STATS, run with BSX_SECTOR_STATS=1): distinct 64-byte sectors the group scan touches per launch
void bsx_sector_report(void);
size_t bsx_hstate_bytes(void);
size_t bsx_htask_bytes(void);
size_t bsx_htaskout_bytes(void);
int bsx_align_occupancy(int paired);
int bsx_hctrl_occupancy(int paired);  // resident blocks of the control kernel per CU
