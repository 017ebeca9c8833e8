// bsx_internal.h — private structures shared by the translation units of libbsx.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/bsx.h"

#define BSX_REF_MARGIN 400u  // words of padding either side of the concatenated reference (reference dbseq.h:15)
#define BSX_SEGLEN 16u
#define BSX_LDS_CHR 512
#define BSX_ROWCAP (BSX_MAXHITS + 1)  // HitArray / PairArray row length (reference align.h:18, pairs.h:22)
#define BSX_HSET_BITS 15
#define BSX_HSET_SLOTS (1u << BSX_HSET_BITS)  // duplicate-suppression hash set, >= 2x the most hits a read can collect
#define BSX_ENTRY_PAD 512  /* zeroed words behind the index entries: k_hscan loads whole 256-entry chunks */
#define BSX_CTX_HEADROOM_DEFAULT (32ull << 30)  /* free device bytes the optional context table must leave (bsx.h: bsx_ref_set_context) */
#define BSX_SORT_TMP 1280            // scratch entries for sorting one class list      // chromosomes whose anchors are staged in LDS by the align kernel

// everything the align kernel needs, passed by value as the kernel argument
struct DevParams {
    int32_t seed_size, index_interval, max_snp_num, max_num_hits, chains, pairend, min_insert, max_insert;
    int32_t report_repeat_hits, randseed, qual_threshold, zero_qual, max_ns, max_readlen, rrbs, n_adapter;
    int32_t digest_len, digest_pos;
    uint32_t seed_bits;
    uint32_t bit_nt_packed;          // bit_nt[i] in byte i
    uint8_t profile_a[16][16];
    char adapter[10][16];            // only the first 15 characters are ever compared (reference align.cpp:414)
    uint8_t adapter_len[10];
    char digest_site[16];
    // reference
    uint32_t n_chr;
    const uint32_t *refcat, *crefcat, *anchor, *chr_size, *rc_offset;
    // plane copy of both strand copies (bsx_planes_build): per 32 nt a {low-bit word, high-bit word} pair; the rc copy starts
    // plane_rc_off bytes behind the forward one
    const uint32_t *refplane;
    uint32_t plane_rc_off;
    // index (CSR)
    const uint32_t *bucket_off, *bucket_nfwd, *entries;
    // per index entry the 32 reference nt left and right of its seed, four packed words (null: not built) — the main kernel's context prefilter
    const uint32_t *ctx;
    // RRBS site table
    const uint32_t *sites, *site_off;
    // RRBS: entries of a bucket grouped by (segment + 16 * direction); rrbs_goff[key * 32 + group] = first entry (null: ungrouped)
    const uint32_t *rrbs_goff;
    // RRBS: site_bin[site_bin_off[c] + (pos >> BSX_SITE_BIN_SHIFT)] = number of sites of chromosome c below that 4 kb bin (null: plain binary search)
    const uint32_t *site_bin, *site_bin_off;
};
#define BSX_SITE_BIN_SHIFT 12

struct Block { uint32_t id, begin, end; };

struct bsx_ref {
    bsx_params P;
    int device = 0;
    uint32_t n_chr = 0;
    uint64_t n_words = 0;
    uint64_t sum_length = 0;
    std::vector<uint32_t> anchor, chr_size, rc_offset;
    std::vector<std::string> names;
    std::vector<Block> blocks;
    // RRBS (reference RefSeq::CCGG_sites / CCGG_index, dbseq.h:106-107)
    std::vector<std::vector<uint32_t>> sites;
    std::vector<std::vector<std::vector<uint32_t>>> ccgg_index;  // [seg][2*chr + strand]
    // device
    uint32_t *d_refcat = nullptr, *d_crefcat = nullptr, *d_anchor = nullptr, *d_chr_size = nullptr, *d_rc_offset = nullptr;
    uint32_t *d_refplane = nullptr;   // plane copy (bsx_planes_build)
    uint32_t plane_rc_off = 0;
    uint32_t *d_bucket_off = nullptr, *d_bucket_nfwd = nullptr, *d_entries = nullptr, *d_ctx = nullptr;
    uint32_t *d_sites = nullptr, *d_site_off = nullptr, *d_rrbs_goff = nullptr, *d_site_bin = nullptr, *d_site_bin_off = nullptr;
    std::vector<uint32_t> rrbs_entries_host;  // RRBS entries in the reference's order (the device copy is grouped, see bsx_index_build_rrbs)
    uint64_t n_entries = 0;
    bool has_index = false;
    // the entries' context table (d_ctx, 16 bytes per entry; bsx_ref_set_context): 0 never, 1 only if ctx_headroom bytes stay free behind it, 2 whenever it can be allocated
    int ctx_mode = 1;
    uint64_t ctx_headroom = BSX_CTX_HEADROOM_DEFAULT, ctx_bytes = 0;
    bool packed_on_device = false;   // Run_ConvertBinseq ran on the device (bsx_pack.hip)
    int n_batches = 0;   // device batches alive on this reference (the context may only be dropped while none can be running)
    uint64_t synth_seed = 0;
};

extern thread_local std::string g_bsx_err;
int bsx_hip_fail(hipError_t e, const char *what, const char *file, int line);
#define HIP_TRY(x)                                                            \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) return bsx_hip_fail(e_, #x, __FILE__, __LINE__); \
    } while (0)

// bsx_refpack.cpp
int bsx_pack_fasta(const bsx_params &P, const char *text, uint64_t n, bsx_ref &r, std::vector<uint32_t> &refcat,
                   std::vector<uint32_t> &crefcat);
// bsx_pack.hip: the same on the device for line-regular FASTA text (WGBS); 1 = not applicable, the caller takes bsx_pack_fasta
int bsx_pack_fasta_device(const bsx_params &P, const char *text, uint64_t n, bsx_ref &r);
// bsx_index.hip
int bsx_planes_build(bsx_ref *r);   // the plane copy of the packed reference (d_refcat / d_crefcat must be filled)
int bsx_index_build_wgbs(bsx_ref *r);
int bsx_index_build_rrbs(bsx_ref *r, const std::vector<uint32_t> &refcat, const std::vector<uint32_t> &crefcat);
// bsx_synth.hip
int bsx_synth_reads_launch(const bsx_ref *r, uint32_t n, uint32_t read_len, int paired, uint64_t seed, uint32_t first_index, uint8_t *d_seq_a,
                           uint8_t *d_seq_b, hipStream_t stream, int kind = 0, uint8_t *d_qual_a = nullptr, uint8_t *d_qual_b = nullptr);
// bsx_align.hip
void bsx_fill_devparams(const bsx_ref *r, DevParams &d);
