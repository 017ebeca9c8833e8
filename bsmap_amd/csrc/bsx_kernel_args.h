// bsx_kernel_args.h — argument block of the align kernel (one struct passed by value).
#pragma once
#include "bsx_internal.h"

struct AlignArgs {
    DevParams P;
    uint32_t n_units, first_index;
    int32_t debug;             // 1: scratch slab per unit (lists kept for inspection) instead of per wave
    uint32_t rowcap;           // capacity of one hit / pair list row (= -w + 64, see DESIGN.md "cap overshoot")
    const uint8_t *seq[2];     // ASCII reads, mate 0 / 1
    const uint64_t *off[2];    // [n_units+1] byte offsets
    const uint8_t *qual[2];    // may be null
    bsx_hit *hits_out;
    bsx_pair *pairs_out;
    bsx_class_counts *cc[2];   // may be null
    uint16_t *npairs_out;      // [n_units][32], may be null
    uint8_t *scratch;
    uint64_t slab_bytes;
    uint32_t *queue;           // work queue head (zeroed before launch)
    uint64_t *counters;        // BSX_N_COUNTERS
    uint8_t *dbg_plan;         // [n_units][128]: start[2][16], order[2][16] for mate a then mate b
};

void bsx_launch_align(const AlignArgs &A, int paired, int grid_blocks, hipStream_t stream);
int bsx_align_occupancy(int paired);
