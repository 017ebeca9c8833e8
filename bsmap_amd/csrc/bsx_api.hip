// bsx_api.hip — C ABI glue of libbsx.so: handles, device memory, uploads/downloads, kernel launches.
// See include/bsx.h for the reference interfaces each entry point stands in for.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <time.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <cmath>
#include <vector>

#include "bsx_internal.h"
#include "bsx_kernel_args.h"

static int g_waves_per_cu = 0;
static int g_heavy_threshold = 0;  // candidate-list length that sends a unit to the heavy pipeline; 0 = by mode (heavy_threshold_for)

extern "C" int bsx_set_waves_per_cu(int w) { g_waves_per_cu = w; return BSX_OK; }
extern "C" int bsx_set_heavy_threshold(int t) { g_heavy_threshold = t < 0 ? 0 : t; return BSX_OK; }
// measured on the BASELINE configs: WGBS is flat from 8 192 to 65 536; RRBS reads of the mid-size repeat families
// cost the main kernel a millisecond each (one wave, 4 MB slab) and go through the scan kernel instead
static uint32_t heavy_threshold_for(const bsx_params &p) { return g_heavy_threshold > 0 ? (uint32_t)g_heavy_threshold : (p.rrbs ? 2048u : 32768u); }
static bool g_user_limits = false;
static uint32_t g_hcap = 24576, g_task_cap = 1048576;  // a 2^20-pair batch defers ~9.4 K units (C3) to ~17.3 K (C5, trimmed reads)
extern "C" int bsx_set_heavy_limits(uint32_t units_per_round, uint32_t task_pool)
{
    if (units_per_round == 0 && task_pool == 0) { g_hcap = 24576; g_task_cap = 1048576; g_user_limits = false; return BSX_OK; }  // back to the defaults
    if (units_per_round < 1 || task_pool < 2 || task_pool > (1u << 22)) return BSX_ERR_ARG;  // (a task record is 8 KB: 2^22 tasks = 34 GB)
    g_hcap = units_per_round; g_task_cap = task_pool; g_user_limits = true;
    return BSX_OK;
}

extern "C" int bsx_device_numa_node(int device)
{
    char id[64] = {0};
    if (hipDeviceGetPCIBusId(id, (int)sizeof id, device) != hipSuccess) return -1;
    for (char *q = id; *q; q++) *q = (char)tolower((unsigned char)*q);
    const std::string path = std::string("/sys/bus/pci/devices/") + id + "/numa_node";
    int node = -1;
    if (FILE *f = fopen(path.c_str(), "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
    return node;
}

extern "C" int bsx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int check_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { g_bsx_err = "no HIP device visible; libbsx has no CPU fallback"; return BSX_ERR_NODEVICE; }
    if (device < 0 || device >= n) { g_bsx_err = "device ordinal out of range"; return BSX_ERR_NODEVICE; }
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_bsx_err = std::string("device is ") + prop.gcnArchName + ", libbsx is built for gfx950 only";
        return BSX_ERR_NODEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    return BSX_OK;
}

template <class T> static int upload(T **dst, const T *src, size_t n)
{
    HIP_TRY(hipMalloc((void **)dst, (n ? n : 1) * sizeof(T)));
    if (n) HIP_TRY(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return BSX_OK;
}

// the per-chromosome tables and the plane copy, behind the packed words (host-packed and uploaded, or packed on the device)
static int finish_ref_tables(bsx_ref *r)
{
    int rc;
    if ((rc = upload(&r->d_anchor, r->anchor.data(), r->anchor.size()))) return rc;
    if ((rc = upload(&r->d_chr_size, r->chr_size.data(), r->chr_size.size()))) return rc;
    if ((rc = upload(&r->d_rc_offset, r->rc_offset.data(), r->rc_offset.size()))) return rc;
    return bsx_planes_build(r);
}

static int finish_ref_upload(bsx_ref *r, const std::vector<uint32_t> &refcat, const std::vector<uint32_t> &crefcat)
{
    // 64 spare words behind each copy so that 16-byte candidate loads never leave the allocation
    // (both strand copies in one allocation, the rc copy right behind the forward one: a candidate's reference words are
    //  then addressed by one 32-bit byte offset whatever its strand — the scan kernel's tail queue relies on that)
    HIP_TRY(hipMalloc((void **)&r->d_refcat, 2 * (r->n_words + 64) * 4));
    r->d_crefcat = r->d_refcat + r->n_words + 64;
    HIP_TRY(hipMemset(r->d_refcat, 0, 2 * (r->n_words + 64) * 4));
    HIP_TRY(hipMemcpy(r->d_refcat, refcat.data(), r->n_words * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(r->d_crefcat, crefcat.data(), r->n_words * 4, hipMemcpyHostToDevice));
    return finish_ref_tables(r);
}

extern "C" int bsx_ref_create_from_fasta(const bsx_params *p, const char *text, uint64_t n_bytes, int device, bsx_ref **out)
{
    if (!p || !text || !out) return BSX_ERR_ARG;
    int rc = check_device(device);
    if (rc) return rc;
    bsx_ref *r = new bsx_ref();
    r->P = *p; r->device = device;
    std::vector<uint32_t> refcat, crefcat;
    // Run_ConvertBinseq on the device where the text allows it (bsx_pack.hip: line-regular FASTA, WGBS): the text goes up as it is and is packed there; any
    // other text — and every RRBS reference, whose site tables come from the text — through the host packer, which applies the reference's token rules
    rc = bsx_pack_fasta_device(*p, text, n_bytes, *r);
    if (rc == BSX_OK) { r->packed_on_device = true; rc = r->n_chr == 0 ? BSX_ERR_IO : finish_ref_tables(r); }
    else if (rc == 1) {
        rc = bsx_pack_fasta(*p, text, n_bytes, *r, refcat, crefcat);
        if (rc == BSX_OK && r->n_chr == 0) rc = BSX_ERR_IO;
        if (rc == BSX_OK) rc = finish_ref_upload(r, refcat, crefcat);
    }
    if (rc == BSX_OK && p->rrbs) rc = bsx_index_build_rrbs(r, refcat, crefcat);
    if (rc == BSX_OK && hipDeviceSynchronize() != hipSuccess) rc = BSX_ERR_DEVICE;  // null-stream memsets must land before any batch stream runs
    if (rc != BSX_OK) { bsx_ref_destroy(r); return rc; }
    *out = r;
    return BSX_OK;
}

extern "C" int bsx_ref_create_from_file(const bsx_params *p, const char *path, int device, bsx_ref **out)
{
    if (!path) return BSX_ERR_ARG;
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) { g_bsx_err = std::string("cannot open ") + path; return BSX_ERR_IO; }  // "fatal error: failed to open ref file" (main.cpp:458)
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size <= 0) { ::close(fd); g_bsx_err = std::string("cannot read ") + path; return BSX_ERR_IO; }
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) { g_bsx_err = std::string("cannot map ") + path; return BSX_ERR_IO; }
    const int rc = bsx_ref_create_from_fasta(p, (const char *)m, (uint64_t)st.st_size, device, out);
    munmap(m, (size_t)st.st_size);
    return rc;
}

extern "C" void bsx_ref_destroy(bsx_ref *r)
{
    if (!r) return;
    (void)hipSetDevice(r->device);
    for (void *q : {(void *)r->d_refcat, (void *)r->d_refplane, (void *)r->d_anchor, (void *)r->d_chr_size, (void *)r->d_rc_offset,
                    (void *)r->d_bucket_off, (void *)r->d_bucket_nfwd, (void *)r->d_entries, (void *)r->d_ctx, (void *)r->d_sites, (void *)r->d_site_off, (void *)r->d_rrbs_goff, (void *)r->d_site_bin, (void *)r->d_site_bin_off})
        if (q) (void)hipFree(q);
    delete r;
}

extern "C" uint32_t bsx_ref_n_chr(const bsx_ref *r) { return r ? r->n_chr : 0; }
extern "C" int bsx_ref_packed_on_device(const bsx_ref *r) { return r && r->packed_on_device ? 1 : 0; }
extern "C" uint64_t bsx_ref_n_words(const bsx_ref *r) { return r ? r->n_words : 0; }
extern "C" uint32_t bsx_ref_n_blocks(const bsx_ref *r) { return r ? (uint32_t)r->blocks.size() : 0; }
extern "C" int bsx_ref_info(const bsx_ref *r, uint32_t *anchor, uint32_t *chr_size, uint32_t *rc_offset)
{
    if (!r) return BSX_ERR_ARG;
    if (anchor) memcpy(anchor, r->anchor.data(), r->anchor.size() * 4);
    if (chr_size) memcpy(chr_size, r->chr_size.data(), r->chr_size.size() * 4);
    if (rc_offset) memcpy(rc_offset, r->rc_offset.data(), r->rc_offset.size() * 4);
    return BSX_OK;
}
extern "C" const char *bsx_ref_chr_name(const bsx_ref *r, uint32_t c) { return (r && c < r->names.size()) ? r->names[c].c_str() : ""; }
extern "C" int bsx_ref_blocks(const bsx_ref *r, uint32_t *id, uint32_t *begin, uint32_t *end)
{
    if (!r) return BSX_ERR_ARG;
    for (size_t i = 0; i < r->blocks.size(); i++) { id[i] = r->blocks[i].id; begin[i] = r->blocks[i].begin; end[i] = r->blocks[i].end; }
    return BSX_OK;
}
extern "C" int bsx_ref_download_words(const bsx_ref *r, uint32_t *refcat, uint32_t *crefcat)
{
    if (!r) return BSX_ERR_ARG;
    HIP_TRY(hipSetDevice(r->device));
    if (refcat) HIP_TRY(hipMemcpy(refcat, r->d_refcat, r->n_words * 4, hipMemcpyDeviceToHost));
    if (crefcat) HIP_TRY(hipMemcpy(crefcat, r->d_crefcat, r->n_words * 4, hipMemcpyDeviceToHost));
    return BSX_OK;
}

extern "C" int bsx_index_build(bsx_ref *r)
{
    if (!r) return BSX_ERR_ARG;
    if (r->P.rrbs) return r->has_index ? BSX_OK : BSX_ERR_STATE;  // RRBS index is assembled with the reference
    return bsx_index_build_wgbs(r);
}
extern "C" uint64_t bsx_index_n_entries(const bsx_ref *r) { return r ? r->n_entries : 0; }

extern "C" int bsx_ref_set_context(bsx_ref *r, int mode, uint64_t headroom_bytes)
{
    if (!r || mode < 0 || mode > 2) return BSX_ERR_ARG;
    r->ctx_mode = mode;
    r->ctx_headroom = headroom_bytes ? headroom_bytes : BSX_CTX_HEADROOM_DEFAULT;
    return BSX_OK;
}
extern "C" uint64_t bsx_ref_context_bytes(const bsx_ref *r) { return r ? r->ctx_bytes : 0; }
extern "C" int bsx_ref_drop_context(bsx_ref *r)
{
    if (!r) return BSX_ERR_ARG;
    if (r->n_batches > 0) { g_bsx_err = "the context table can only be dropped while no batch of the reference exists"; return BSX_ERR_STATE; }
    if (r->d_ctx) { HIP_TRY(hipSetDevice(r->device)); (void)hipFree(r->d_ctx); r->d_ctx = nullptr; r->ctx_bytes = 0; }
    return BSX_OK;
}
extern "C" int bsx_index_download(const bsx_ref *r, uint32_t *bucket_off, uint32_t *bucket_nfwd, uint32_t *entries)
{
    if (!r) return BSX_ERR_ARG;
    if (!r->has_index) return BSX_ERR_STATE;
    HIP_TRY(hipSetDevice(r->device));
    const size_t K = r->P.total_kmers;
    if (bucket_off) HIP_TRY(hipMemcpy(bucket_off, r->d_bucket_off, (K + 1) * 4, hipMemcpyDeviceToHost));
    if (bucket_nfwd) HIP_TRY(hipMemcpy(bucket_nfwd, r->d_bucket_nfwd, K * 4, hipMemcpyDeviceToHost));
    if (entries && r->n_entries) {
        if (r->P.rrbs && !r->rrbs_entries_host.empty()) memcpy(entries, r->rrbs_entries_host.data(), r->n_entries * 8);  // the reference's order (the device copy is grouped)
        else HIP_TRY(hipMemcpy(entries, r->d_entries, r->n_entries * (r->P.rrbs ? 8 : 4), hipMemcpyDeviceToHost));
    }
    return BSX_OK;
}
extern "C" uint32_t bsx_ref_n_sites(const bsx_ref *r, uint32_t c) { return (r && c < r->sites.size()) ? (uint32_t)r->sites[c].size() : 0; }
extern "C" int bsx_ref_sites(const bsx_ref *r, uint32_t c, uint32_t *sites)
{
    if (!r || c >= r->sites.size()) return BSX_ERR_ARG;
    memcpy(sites, r->sites[c].data(), r->sites[c].size() * 4);
    return BSX_OK;
}

void bsx_fill_devparams(const bsx_ref *r, DevParams &d)
{
    const bsx_params &p = r->P;
    memset(&d, 0, sizeof(d));
    d.seed_size = p.seed_size; d.index_interval = p.index_interval; d.max_snp_num = p.max_snp_num; d.max_num_hits = p.max_num_hits;
    d.chains = p.chains; d.pairend = p.pairend; d.min_insert = p.min_insert; d.max_insert = p.max_insert;
    d.report_repeat_hits = p.report_repeat_hits; d.randseed = p.randseed; d.qual_threshold = p.qual_threshold; d.zero_qual = p.zero_qual;
    d.max_ns = p.max_ns; d.max_readlen = p.max_readlen; d.rrbs = p.rrbs; d.n_adapter = p.n_adapter;
    d.digest_len = (int)strlen(p.digest_site); d.digest_pos = p.digest_pos;
    d.seed_bits = p.seed_bits;
    d.bit_nt_packed = p.bit_nt[0] | (p.bit_nt[1] << 8) | (p.bit_nt[2] << 16) | ((uint32_t)p.bit_nt[3] << 24);
    memcpy(d.profile_a, p.profile_a, sizeof(d.profile_a));
    for (int i = 0; i < p.n_adapter; i++) {
        size_t l = strlen(p.adapter[i]);
        d.adapter_len[i] = (uint8_t)(l > 15 ? 15 : l);
        memcpy(d.adapter[i], p.adapter[i], d.adapter_len[i]);
    }
    memcpy(d.digest_site, p.digest_site, 16);
    d.n_chr = r->n_chr;
    d.refplane = r->d_refplane; d.plane_rc_off = r->plane_rc_off;
    d.refcat = r->d_refcat; d.crefcat = r->d_crefcat; d.anchor = r->d_anchor; d.chr_size = r->d_chr_size; d.rc_offset = r->d_rc_offset;
    d.bucket_off = r->d_bucket_off; d.bucket_nfwd = r->d_bucket_nfwd; d.entries = r->d_entries; d.ctx = r->d_ctx;
    d.sites = r->d_sites; d.site_off = r->d_site_off; d.rrbs_goff = r->d_rrbs_goff; d.site_bin = r->d_site_bin; d.site_bin_off = r->d_site_bin_off;
}

// ---------------------------------------------------------------------------------------------------------------
// batches
// ---------------------------------------------------------------------------------------------------------------
#define BSX_MAX_GROUPS 8
#define BSX_POLL_SLOTS 4
struct bsx_batch {
    bsx_ref *ref = nullptr;
    int paired = 0, debug = 0, has_qual = 0, leak_exact = 0, work_counters = 1;
    uint32_t n_hist = 0;
    uint8_t *d_hist_seq[2] = {nullptr, nullptr}, *d_hist_qual[2] = {nullptr, nullptr};
    uint64_t *d_hist_off[2] = {nullptr, nullptr};
    uint8_t *d_leak_rec = nullptr;
    // exact mode: per-position records of the mate streams (k_leak_meta), block summaries, the state before / behind the stream
    uint16_t *d_leak_meta[2] = {nullptr, nullptr};
    uint32_t *d_leak_blk = nullptr;   // [4][n_blk]: blkmax mate 0 / 1, blkset mate 0 / 1
    uint32_t leak_nblk = 0;
    uint8_t *d_leak_init = nullptr, *d_leak_final = nullptr;
    bool leak_meta_valid = false, leak_has_init = false;
    uint32_t max_units = 0, n_units = 0, first_index = 0;
    hipStream_t stream = nullptr;
    // heavy pipeline: unit groups whose passes run out of phase — the scan passes of all groups on `stream`, the control passes of
    // group g on its own high-priority stream, ordered against each other by events on the device (no host read-back per pass)
    struct Group {
        hipStream_t s_ctrl = nullptr;
        hipEvent_t ev_ctrl = nullptr, ev_scan = nullptr, ev_poll[BSX_POLL_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
        uint32_t *d_bins = nullptr, *d_bstart = nullptr, *d_chunk_tot = nullptr, *d_rank = nullptr, *d_order = nullptr, *d_glist = nullptr;
    } grp[BSX_MAX_GROUPS];
    int n_groups = 1, chunk_passes = 2, trace = 0, hctrl_blocks_per_cu = 1;
    uint32_t tail_tasks = 16384, tail_grid_tasks = 4096;  // a group whose passes publish fewer tasks than tail_tasks scans them with a grid for tail_grid_tasks on its control stream
    uint32_t bin_shift = 0, n_bins = 1;
    hipEvent_t ev_sync = nullptr, ev_wait = nullptr;
    std::vector<hipEvent_t> scan_ev;  // pairs of timing events around every k_hscan launch of the last run (pool grows on demand)
    size_t scan_ev_used = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint8_t *d_seq[2] = {nullptr, nullptr}, *d_qual[2] = {nullptr, nullptr};
    uint64_t *d_off[2] = {nullptr, nullptr};
    size_t seq_cap[2] = {0, 0};
    bsx_hit *d_hits = nullptr;
    bsx_pair *d_pairs = nullptr;
    bsx_class_counts *d_cc[2] = {nullptr, nullptr};
    uint16_t *d_npairs = nullptr;
    uint8_t *d_scratch = nullptr, *d_dbg = nullptr;
    uint32_t *d_cycles = nullptr;
    uint32_t *d_heavy_list = nullptr, *d_heavy_count = nullptr;
    // heavy pipeline pools
    uint8_t *d_hstate = nullptr, *d_hslabs = nullptr, *d_htasks = nullptr, *d_htout = nullptr;
    uint32_t *d_hactive[2] = {nullptr, nullptr}, *d_hcnt = nullptr;  // d_hcnt: per group two ping-pong counter blocks (BSX_HCNT_*: active units, tasks, queue head)
    uint32_t hcap = 0, task_cap = 0;
    bool sig_hist = false;    // diagnostics: histogram of tasks per identical window (bsx_sig_hist_pass)
    bool same_kernel = false;    // the scan kernel is k_hscan_same (WGBS unless BSX_SAME=0, RRBS with BSX_SAME=2): read once, at creation
    bool sector_stats = false;   // diagnostics (a build with -DBSX_SECTOR_STATS): distinct sectors per scan launch
    uint32_t xcd_map = 128;   // order_block (bsx_align.hip): pieces of 128 scan blocks dealt to the XCDs in turn
    uint32_t *h_pinned = nullptr;  // pinned host words for the per-pass count read-backs
    int n_cu = 0;
    uint32_t last_heavy = 0, last_heavy_iters = 0, last_redo = 0;
    size_t scratch_bytes = 0;
    uint32_t *d_queue = nullptr;
    uint64_t *d_counters = nullptr, *d_scan_stats = nullptr;
    uint64_t slab_bytes = 0, hslab_bytes = 0;
    uint32_t *d_redo = nullptr;  // [max_units + 1]: count, then unit ids
    uint32_t rowcap = 0, hkcap = 0;  // hkcap: key capacity of a deferred unit's duplicate set (BSX_HEAVY_KCAP test hook, read at creation)
    int grid_blocks = 0;
    std::vector<hipEvent_t> ctrl_ev;   // per control pass: before k_hctrl, behind it, behind the order kernels (bsx_batch_stage_ms)
    size_t ctrl_ev_used = 0;
    hipEvent_t ev_align = nullptr;     // behind the main kernel (and the exact mode's pre-pass)
    bool stage_timing = false;         // bsx_batch_set_stage_timing: the events above are only recorded on request (three more per control pass on its stream)
    bool ran = false;
    bool counted = false;        // the batch is in its reference's n_batches
};

// Waiting without burning a CPU.  hipStreamSynchronize busy-waits, and so — measured on this ROCm (tools/driver_cpu.py: thread CPU time =
// wall time of a Do_Batch) — does hipEventSynchronize on an event created with hipEventBlockingSync.  The thread that drives a batch
// therefore polls the event and sleeps in between: 50 us at first (short waits stay short), 200 us after that.  The waits that sit on a
// batch's critical path are two (the deferred-unit count behind the main kernel, the end of the run); the per-chunk polls of the heavy
// pipeline run two chunks behind the queue.  Eight GPUs x three batches are 24 such threads under one CPU quota.
static hipError_t wait_event(hipEvent_t ev)
{
    for (int n = 0;; n++) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return hipSuccess;
        if (e != hipErrorNotReady) return e;
        (void)hipGetLastError();  // (hipErrorNotReady is sticky for hipGetLastError)
        struct timespec ts = {0, n < 8 ? 50000L : 200000L};
        nanosleep(&ts, nullptr);
    }
}
static hipError_t stream_wait(bsx_batch *b);

// duplicate-suppression set of one mate slab (Slab in bsx_align.hip).  WGBS and paired RRBS: every remembered coordinate is a
// hit, the -w caps bound them.  Single-end RRBS also remembers the coordinates its fragment-size filter rejects
// (align.cpp:201-207), which nothing caps: a poly-T read of the hg38-sized genome collects thousands — 2^18 keys, 2^19 slots.
// (BSX_KCAP: test hook that makes the large set of single-end RRBS small, to exercise the BSX_F_LIMIT path)
static uint32_t key_cap(const bsx_params &p, uint32_t rowcap) { return (p.rrbs && !p.pairend) ? (getenv("BSX_KCAP") ? (uint32_t)std::max(64, atoi(getenv("BSX_KCAP"))) : (1u << 18)) : (uint32_t)(p.max_snp_num + 2) * rowcap; }
static uint32_t hset_bits(const bsx_params &p) { return (p.rrbs && !p.pairend) ? 19u : (uint32_t)BSX_HSET_BITS; }

// the small set of the heavy pipeline's slabs (BSX_HEAVY_KCAP: test hook that makes it overflow)
static uint32_t heavy_key_cap(const bsx_params &p, uint32_t rowcap) { return getenv("BSX_HEAVY_KCAP") ? (uint32_t)std::max(64, atoi(getenv("BSX_HEAVY_KCAP"))) : (uint32_t)(p.max_snp_num + 2) * rowcap; }

static uint64_t mate_bytes(const bsx_params &p, uint32_t rowcap, bool heavy = false)
{
    const uint64_t rows = (uint64_t)p.max_snp_num + 2;  // nclass + 1 spare row (see Slab in bsx_align.hip)
    const uint64_t kc = heavy ? heavy_key_cap(p, rowcap) : key_cap(p, rowcap), hb = heavy ? (uint64_t)BSX_HSET_BITS : hset_bits(p);
    return 2 * rows * rowcap * 8 + 2 * kc * 4 + ((uint64_t)4 << hb) + (uint64_t)BSX_SORT_TMP * 8;
}

static uint64_t slab_size(const bsx_params &p, int paired, uint32_t rowcap, bool heavy = false)
{
    const uint64_t mate = mate_bytes(p, rowcap, heavy);
    uint64_t s = mate;
    if (paired) s = 2 * mate + (2 * (uint64_t)p.max_snp_num + 2) * rowcap * 24;
    return (s + 255) & ~255ull;
}

static hipError_t stream_wait(bsx_batch *b)
{
    if (!b->ev_wait) return hipStreamSynchronize(b->stream);
    hipError_t e = hipEventRecord(b->ev_wait, b->stream);
    if (e != hipSuccess) return e;
    return wait_event(b->ev_wait);
}

// What a device batch will allocate, computed on the host from its parameters alone (bsx_batch_plan_bytes: bench.py and the CPU suite
// check a run's whole plan against the device's memory before anything is allocated; ensure_scratch allocates exactly this).
struct BatchPlan {
    uint64_t per_unit_bytes = 0;   // reads, offsets, result records, deferred / redo lists: follows max_units
    uint64_t scratch_bytes = 0;    // the main kernel's per-wave slabs: follows the grid, cannot shrink
    uint64_t pool_bytes = 0;       // work pools of the heavy pipeline at their starting size (halved until they fit)
    uint32_t hcap = 0, task_cap = 0, n_bins = 1, bin_shift = 0;
    int grid_blocks = 0;
};
static uint64_t pool_bytes_for(const bsx_params &P, int paired, uint32_t hcap, uint32_t task_cap, uint32_t n_bins, int n_groups, uint64_t hslab_bytes)
{
    (void)P; (void)paired;
    const uint64_t tcap = task_cap / (uint32_t)n_groups;
    return (uint64_t)hcap * (bsx_hstate_bytes() + hslab_bytes + 8) + (uint64_t)task_cap * (bsx_htask_bytes() + bsx_htaskout_bytes()) +
           (uint64_t)n_groups * ((uint64_t)n_bins * 8 + (uint64_t)bsx_bin_chunks(n_bins) * 4 + tcap * 12 + 16);
}
// starting sizes of the heavy pipeline's pools for runs of `units` units (bsx_set_heavy_limits replaces them)
static void default_pools(const bsx_params &P, uint64_t hslab, uint32_t units, uint32_t &hcap, uint32_t &task_cap)
{
    // Round 5: the pools follow the batch beyond 2^21 units.  The tasks of one window AND read offset that a pass holds are what k_hscan_same
    // evaluates as a group, and their number grows with the deferred units of the pass: C3 at 2^22 pairs per batch with one round of 40 K units
    // and 2 M tasks 27.0 M reads/s, with the 2^20-pair pools (two rounds of 19 K) 25.4 M, at 2^20 pairs per batch 24.3 M (profiles/r05d).
    // WGBS: one deferred unit per 100 units (C3 defers 0.9 %), up to 64 GB of slabs; tasks: half a task per unit, up to 2^22.
    const uint64_t slab_budget = P.rrbs ? (40ull << 30) : (26ull << 30);
    uint64_t hc = std::min<uint64_t>(262144, std::max<uint64_t>(24576, slab_budget / hslab));
    if (!P.rrbs) hc = std::max<uint64_t>(hc, std::min<uint64_t>((uint64_t)units / 100, (64ull << 30) / hslab));
    hcap = (uint32_t)std::min<uint64_t>(units, hc);
    uint32_t task_default = P.rrbs ? 2u * 1048576u : 1048576u;
    if (!P.rrbs) task_default = (uint32_t)std::min<uint64_t>(1ull << 22, std::max<uint64_t>(task_default, (uint64_t)units / 2));
    task_cap = std::min<uint32_t>(task_default, std::max<uint32_t>(4096u, 64u * hcap));
}
extern "C" int bsx_default_heavy_limits(const bsx_params *p, uint32_t units, int paired, uint32_t *units_per_round, uint32_t *task_pool)
{
    if (!p || !units || !units_per_round || !task_pool) return BSX_ERR_ARG;
    default_pools(*p, slab_size(*p, paired ? 1 : 0, BSX_ROWCAP, true), units, *units_per_round, *task_pool);
    return BSX_OK;
}

static BatchPlan plan_batch(const bsx_params &P, int paired, uint32_t max_units, uint64_t n_entries, int n_cu, int blocks_per_cu, int n_groups, bool debug)
{
    BatchPlan pl;
    const uint32_t rowcap = BSX_ROWCAP;
    const uint64_t slab = slab_size(P, paired, rowcap), hslab = slab_size(P, paired, rowcap, true);
    int grid = n_cu * std::min(8, std::max(1, blocks_per_cu));
    const int need = (int)((max_units + 3) / 4);
    if (grid > need) grid = need > 0 ? need : 1;
    pl.grid_blocks = grid;
    pl.scratch_bytes = (debug ? (uint64_t)max_units : (uint64_t)grid * 4) * slab;
    const int nm = paired ? 2 : 1;
    pl.per_unit_bytes = (uint64_t)nm * (2 * ((uint64_t)max_units * 160 + 256) + ((uint64_t)max_units + 1) * 8 + (uint64_t)max_units * sizeof(bsx_class_counts)) +
                        (paired ? (uint64_t)max_units * (sizeof(bsx_pair) + 64) : (uint64_t)max_units * sizeof(bsx_hit)) + 2 * ((uint64_t)max_units + 1) * 4;
    // deferred units handled per round (more than this: several rounds): what 26 GB of slabs hold — 24 576 units of the 1.07 MB paired
    // -v 6 slab, 111 K of the 234 KB single-end -v 2 one.  RRBS defers a third of its reads (Alu-like fragments) and its scan kernel
    // shares work between the reads that walk one window in the same pass: the more units a round holds, the longer those runs — 40 GB
    // of slabs (170 K units) and 2 M tasks: C4 195 -> 159 ms per step.  Task records (8 KB each): 1 M for WGBS (C5 358 -> 328 ms against
    // 512 K: fewer requests refused), following the batch size for small batches.  bsx_set_heavy_limits replaces the STARTING sizes;
    // either way the pools are halved until they fit (ensure_scratch).
    if (g_user_limits) { pl.hcap = std::min<uint32_t>(max_units, g_hcap); pl.task_cap = g_task_cap; }
    else default_pools(P, hslab, max_units, pl.hcap, pl.task_cap);
    // scan order of a pass: bins of 2^shift index entries, at most 2^20 of them (bsx_launch_task_order)
    // (WGBS: 2^21 — the tasks of one window are dealt over the bins it covers by read offset, for k_hscan_same: 8 bins of 1 024 entries per window at hg38 size)
    const uint64_t ne = std::max<uint64_t>(1, n_entries);
    const uint32_t bin_log2 = getenv("BSX_BIN_LOG2") ? (uint32_t)std::max(8, std::min(21, atoi(getenv("BSX_BIN_LOG2")))) : (P.rrbs ? 20u : 21u);  // tuning knob
    while (((ne >> pl.bin_shift) + 1) > (1u << bin_log2)) pl.bin_shift++;
    pl.n_bins = (uint32_t)(ne >> pl.bin_shift) + 1;
    pl.pool_bytes = pool_bytes_for(P, paired, pl.hcap, pl.task_cap, pl.n_bins, n_groups, hslab);
    return pl;
}

// Device memory a batch keeps free behind its own allocations.  Pools that took everything but a flat 4 GB left no room for the NEXT batch's
// fixed part (its per-wave slabs alone are 5.5 GB for paired -v 6, 17 GB for single-end RRBS): the driver's bench command died there in round 4.
// A batch therefore leaves room for one more batch like itself (slabs + per-unit arrays) plus 4 GB; bsx_set_pool_reserve overrides.
static uint64_t g_pool_reserve = 0;  // 0 = by batch
extern "C" int bsx_set_pool_reserve(uint64_t bytes) { g_pool_reserve = bytes; return BSX_OK; }

extern "C" int bsx_batch_plan_bytes(const bsx_params *p, uint32_t max_units, int paired, uint64_t n_entries, uint32_t n_cu, uint32_t blocks_per_cu, uint64_t *out3)
{
    if (!p || !out3 || max_units == 0 || n_cu == 0) return BSX_ERR_ARG;
    int ng = 1;
    if (const char *e = getenv("BSX_HEAVY_GROUPS")) ng = std::max(1, std::min(BSX_MAX_GROUPS, atoi(e)));
    const BatchPlan pl = plan_batch(*p, paired ? 1 : 0, max_units, n_entries, (int)n_cu, (int)blocks_per_cu, ng, false);
    out3[0] = pl.per_unit_bytes; out3[1] = pl.scratch_bytes; out3[2] = pl.pool_bytes;
    return BSX_OK;
}

static int ensure_scratch(bsx_batch *b)
{
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, b->ref->device));
    int blocks_per_cu = bsx_align_occupancy(b->paired);
    if (g_waves_per_cu > 0) blocks_per_cu = std::max(1, std::min(blocks_per_cu, g_waves_per_cu / 4));
    if (blocks_per_cu > 8) blocks_per_cu = 8;
    int grid = prop.multiProcessorCount * blocks_per_cu;
    const int need = (int)((b->max_units + 3) / 4);
    if (grid > need) grid = need > 0 ? need : 1;
    b->grid_blocks = grid;
    b->n_cu = prop.multiProcessorCount;
    // control blocks per CU: one, although two fit.  A control wave holds half of its SIMD's registers: with two blocks per CU the scan
    // kernel's waves cannot be resident beside them and the two only alternate; with one, four scan waves per SIMD run beside it
    // (C3 with two batches in flight 136.5 -> 131.3 ms per step, C4 211 -> 195; BSX_HCTRL_BLOCKS: tuning knob)
    b->hctrl_blocks_per_cu = 1;
    if (const char *e = getenv("BSX_HCTRL_BLOCKS")) b->hctrl_blocks_per_cu = std::max(1, std::min(bsx_hctrl_occupancy(b->paired), atoi(e)));
    if (const char *e = getenv("BSX_TAIL_TASKS")) b->tail_tasks = (uint32_t)std::max(0, atoi(e));           // 0: no tail mode (tuning)
    if (const char *e = getenv("BSX_TAIL_GRID")) b->tail_grid_tasks = (uint32_t)std::max(64, atoi(e));
    // (the main kernel's per-wave slabs first: what cannot shrink; the pools below take what is left)
    const uint64_t slots = b->debug ? b->max_units : (uint64_t)grid * 4;
    const size_t bytes = (size_t)(slots * b->slab_bytes);
    if (bytes > b->scratch_bytes) {
        if (b->d_scratch) (void)hipFree(b->d_scratch);
        b->d_scratch = nullptr;
        HIP_TRY(hipMalloc((void **)&b->d_scratch, bytes));
        HIP_TRY(hipMemsetAsync(b->d_scratch, 0, bytes, b->stream));  // the per-slab hash sets must start empty (same stream as the kernels)
        b->scratch_bytes = bytes;
    }
    if (!b->d_heavy_list) {
        const BatchPlan pl = plan_batch(b->ref->P, b->paired, b->max_units, b->ref->n_entries, prop.multiProcessorCount, blocks_per_cu, b->n_groups, b->debug != 0);
        b->hcap = pl.hcap; b->task_cap = pl.task_cap;
        HIP_TRY(hipMalloc((void **)&b->d_heavy_list, ((size_t)b->max_units + 1) * 4));
        HIP_TRY(hipMalloc((void **)&b->d_heavy_count, 256));
        HIP_TRY(hipHostMalloc((void **)&b->h_pinned, 1024, hipHostMallocDefault));
        HIP_TRY(hipMalloc((void **)&b->d_redo, ((size_t)b->max_units + 1) * 4));
        HIP_TRY(hipMalloc((void **)&b->d_hcnt, (size_t)BSX_MAX_GROUPS * 2 * BSX_HCNT_BLOCK * 4));
        b->bin_shift = pl.bin_shift; b->n_bins = pl.n_bins;
        // The pools proper.  Their default sizes are for a device that holds two or three batches (31 GB each for WGBS, 74 GB for RRBS);
        // where that much is not free — more batches per device, a smaller or shared device — the pools are halved until they fit
        // (more rounds and refused requests then, same results) instead of failing the batch.
        auto free_pools = [&]() {
            void **ptrs[] = {(void **)&b->d_hstate, (void **)&b->d_hslabs, (void **)&b->d_htasks, (void **)&b->d_htout, (void **)&b->d_hactive[0], (void **)&b->d_hactive[1]};
            for (void **q : ptrs) { if (*q) (void)hipFree(*q); *q = nullptr; }
            for (int g = 0; g < b->n_groups; g++) {
                bsx_batch::Group &q = b->grp[g];
                void **gp[] = {(void **)&q.d_bins, (void **)&q.d_bstart, (void **)&q.d_chunk_tot, (void **)&q.d_rank, (void **)&q.d_order, (void **)&q.d_glist};
                for (void **x : gp) { if (*x) (void)hipFree(*x); *x = nullptr; }
            }
        };
        // (the group scan kernel: WGBS unless BSX_SAME=0, RRBS with BSX_SAME=2 — the same rule as bsx_batch_run_range)
        const int same_env0 = getenv("BSX_SAME") ? atoi(getenv("BSX_SAME")) : 1;
        const bool same_kernel = b->same_kernel = b->ref->P.rrbs ? same_env0 == 2 : same_env0 != 0;
        auto alloc_pools = [&]() -> hipError_t {
            hipError_t e;
#define POOL_TRY(x) do { if ((e = (x)) != hipSuccess) return e; } while (0)
            POOL_TRY(hipMalloc((void **)&b->d_hstate, (size_t)b->hcap * bsx_hstate_bytes()));
            POOL_TRY(hipMalloc((void **)&b->d_hslabs, (size_t)b->hcap * b->hslab_bytes));
            POOL_TRY(hipMalloc((void **)&b->d_htasks, (size_t)b->task_cap * bsx_htask_bytes()));
            POOL_TRY(hipMalloc((void **)&b->d_htout, (size_t)b->task_cap * bsx_htaskout_bytes()));
            for (int k = 0; k < 2; k++) POOL_TRY(hipMalloc((void **)&b->d_hactive[k], (size_t)b->hcap * 4));
            const uint32_t tcap = b->task_cap / (uint32_t)b->n_groups;
            for (int g = 0; g < b->n_groups; g++) {
                bsx_batch::Group &q = b->grp[g];
                POOL_TRY(hipMalloc((void **)&q.d_bins, (size_t)b->n_bins * 4));
                POOL_TRY(hipMalloc((void **)&q.d_bstart, (size_t)b->n_bins * 4));
                POOL_TRY(hipMalloc((void **)&q.d_chunk_tot, (size_t)bsx_bin_chunks(b->n_bins) * 4));
                POOL_TRY(hipMalloc((void **)&q.d_rank, (size_t)tcap * 4));
                POOL_TRY(hipMalloc((void **)&q.d_order, (size_t)tcap * 4));
                if (same_kernel) POOL_TRY(hipMalloc((void **)&q.d_glist, ((size_t)tcap + 4) * 4));   // (k_hscan_same: start slots of the groups, then their count)
            }
#undef POOL_TRY
            return hipSuccess;
        };
        // Starting sizes (the defaults above, or bsx_set_heavy_limits) are halved until the pools fit with `reserve` bytes to spare: first by
        // arithmetic against hipMemGetInfo, then — another process or thread may be allocating at the same moment — on a failed hipMalloc.
        const uint64_t reserve = g_pool_reserve ? g_pool_reserve : ((4ull << 30) + pl.scratch_bytes + pl.per_unit_bytes);
        for (int attempt = 0;; attempt++) {
            size_t fr = 0, tot = 0;
            const bool smallest = b->hcap <= std::min<uint32_t>(1024u, b->max_units) && b->task_cap <= 4096;
            hipError_t e = hipErrorOutOfMemory;
            if (smallest || hipMemGetInfo(&fr, &tot) != hipSuccess ||
                pool_bytes_for(b->ref->P, b->paired, b->hcap, b->task_cap, b->n_bins, b->n_groups, b->hslab_bytes) + reserve <= fr) {
                e = alloc_pools();
                if (e == hipSuccess) break;
                free_pools();
            }
            if (e != hipErrorOutOfMemory || attempt == 12 || smallest) return bsx_hip_fail(e, "hipMalloc (work pools of the heavy pipeline)", __FILE__, __LINE__);
            (void)hipGetLastError();
            // (the floors never RAISE a pool: a caller's small limits — the tests' way into the many-round paths — stay as they are)
            b->hcap = std::min<uint32_t>(b->hcap, std::max<uint32_t>(std::min<uint32_t>(1024u, b->max_units), b->hcap / 2));
            b->task_cap = std::min<uint32_t>(b->task_cap, std::max<uint32_t>(4096u, b->task_cap / 2));
        }
        if (b->trace) fprintf(stderr, "[bsx] batch %p: pools for %u deferred units per round, %u scan tasks (planned %u / %u)\n", (void *)b, b->hcap, b->task_cap, pl.hcap, pl.task_cap);
        if (getenv("BSX_POISON")) HIP_TRY(hipMemsetAsync(b->d_hstate, 0xA5, (size_t)b->hcap * bsx_hstate_bytes(), b->stream));  // test hook: recycled memory is not zero
        HIP_TRY(hipMemsetAsync(b->d_hslabs, 0, (size_t)b->hcap * b->hslab_bytes, b->stream));
        if (getenv("BSX_POISON")) {
            HIP_TRY(hipMemsetAsync(b->d_htout, 0xA5, (size_t)b->task_cap * bsx_htaskout_bytes(), b->stream));
            HIP_TRY(hipMemsetAsync(b->d_htasks, 0xA5, (size_t)b->task_cap * bsx_htask_bytes(), b->stream));
        }
        for (int g = 0; g < b->n_groups; g++) HIP_TRY(hipMemsetAsync(b->grp[g].d_bins, 0, (size_t)b->n_bins * 4, b->stream));
    }
    return BSX_OK;
}

static int batch_create_once(bsx_ref *r, uint32_t max_units, int paired, bsx_batch **out);
extern "C" int bsx_batch_create(bsx_ref *r, uint32_t max_units, int paired, bsx_batch **out)
{
    int rc = batch_create_once(r, max_units, paired, out);
    // The context table is an accelerator, not a requirement: where the FIXED part of the first batch (per-wave slabs, per-unit arrays, smallest pools) does
    // not fit beside it, it goes and the batch is tried once more (no batch exists, so no kernel can hold its address; later batches find the memory as it is).
    if (rc == BSX_ERR_NOMEM && r && r->d_ctx && r->n_batches == 0 && r->ctx_mode != 2) {
        const std::string first = g_bsx_err;
        (void)hipGetLastError();
        if (bsx_ref_drop_context(r) == BSX_OK) {
            rc = batch_create_once(r, max_units, paired, out);
            if (rc == BSX_OK && getenv("BSX_TRACE_HEAVY")) fprintf(stderr, "[bsx] the index's context table was dropped to make room for the batch (%s)\n", first.c_str());
        }
    }
    if (rc == BSX_OK) r->n_batches++;
    return rc;
}
static int batch_create_once(bsx_ref *r, uint32_t max_units, int paired, bsx_batch **out)
{
    if (!r || !out || max_units == 0) return BSX_ERR_ARG;
    if (!r->has_index) { g_bsx_err = "bsx_index_build must run before bsx_batch_create"; return BSX_ERR_STATE; }
    HIP_TRY(hipSetDevice(r->device));
    bsx_batch *b = new bsx_batch();
    b->ref = r; b->paired = paired ? 1 : 0; b->max_units = max_units;
    b->rowcap = BSX_ROWCAP;
    b->slab_bytes = slab_size(r->P, b->paired, b->rowcap);
    b->hslab_bytes = slab_size(r->P, b->paired, b->rowcap, true);
    b->hkcap = heavy_key_cap(r->P, b->rowcap);
    int rc = BSX_OK;
    auto fail = [&](int code) { bsx_batch_destroy(b); return code; };
    // tuning / diagnostic knobs are read once, here (a batch keeps its settings): unit groups of the heavy pipeline (default 1: the
    // control and scan passes of a batch alternate and overlap those of the OTHER batch in flight — measured best with two batches in
    // flight, C3 131 ms per step against 148 with two groups each; a caller that keeps a single batch in flight gains from 2: 158 against
    // 177 ms), passes enqueued per host poll
    if (const char *e = getenv("BSX_HEAVY_GROUPS")) b->n_groups = std::max(1, std::min(BSX_MAX_GROUPS, atoi(e)));
    if (const char *e = getenv("BSX_XCD_MAP")) b->xcd_map = (uint32_t)std::max(0, atoi(e));   // how k_hscan's blocks map onto the scan order: 0 as dispatched, 1 one contiguous eighth per XCD, N >= 2 pieces of N blocks dealt to the XCDs in turn
    if (const char *e = getenv("BSX_HEAVY_CHUNK")) b->chunk_passes = std::max(1, std::min(64, atoi(e)));
    b->trace = getenv("BSX_TRACE_HEAVY") != nullptr;
    if (const char *e = getenv("BSX_WORK_COUNTERS")) b->work_counters = atoi(e) != 0;   // the default of bsx_batch_set_work_counters (test hook)
    b->sig_hist = getenv("BSX_SIGHIST") != nullptr;
    b->sector_stats = getenv("BSX_SECTOR_STATS") != nullptr;
    if (b->sector_stats) bsx_sector_pass(r, nullptr);   // (sets the bitmap up)
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) return fail(BSX_ERR_DEVICE);
    {
        int lo_p = 0, hi_p = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo_p, &hi_p);
        for (int g = 0; g < b->n_groups; g++) {
            bsx_batch::Group &q = b->grp[g];
            if (hipStreamCreateWithPriority(&q.s_ctrl, hipStreamNonBlocking, hi_p) != hipSuccess) return fail(BSX_ERR_DEVICE);
            for (hipEvent_t *e : {&q.ev_ctrl, &q.ev_scan})
                if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) return fail(BSX_ERR_DEVICE);
            for (int k = 0; k < BSX_POLL_SLOTS; k++)  // the host polls these and sleeps in between (wait_event: one driver thread per device batch)
                if (hipEventCreateWithFlags(&q.ev_poll[k], hipEventDisableTiming | hipEventBlockingSync) != hipSuccess) return fail(BSX_ERR_DEVICE);
        }
        if (hipEventCreateWithFlags(&b->ev_sync, hipEventDisableTiming) != hipSuccess) return fail(BSX_ERR_DEVICE);
        if (hipEventCreateWithFlags(&b->ev_wait, hipEventDisableTiming | hipEventBlockingSync) != hipSuccess) return fail(BSX_ERR_DEVICE);
    }
    if (hipEventCreate(&b->ev0) != hipSuccess || hipEventCreate(&b->ev1) != hipSuccess) return fail(BSX_ERR_DEVICE);
    const int nm = b->paired ? 2 : 1;
    for (int m = 0; m < nm; m++) {
        b->seq_cap[m] = (size_t)max_units * 160 + 256;
        if (hipMalloc((void **)&b->d_seq[m], b->seq_cap[m]) != hipSuccess || hipMalloc((void **)&b->d_qual[m], b->seq_cap[m]) != hipSuccess ||
            hipMalloc((void **)&b->d_off[m], ((size_t)max_units + 1) * 8) != hipSuccess || hipMalloc((void **)&b->d_cc[m], (size_t)max_units * sizeof(bsx_class_counts)) != hipSuccess)
            return fail(BSX_ERR_NOMEM);
    }
    if (b->paired) {
        if (hipMalloc((void **)&b->d_pairs, (size_t)max_units * sizeof(bsx_pair)) != hipSuccess || hipMalloc((void **)&b->d_npairs, (size_t)max_units * 64) != hipSuccess)
            return fail(BSX_ERR_NOMEM);
    } else if (hipMalloc((void **)&b->d_hits, (size_t)max_units * sizeof(bsx_hit)) != hipSuccess) return fail(BSX_ERR_NOMEM);
    if (hipMalloc((void **)&b->d_queue, 256) != hipSuccess || hipMalloc((void **)&b->d_counters, 24 * 8 + 256) != hipSuccess) return fail(BSX_ERR_NOMEM);
    static_assert(BSX_N_COUNTERS <= 24, "the diagnostic clocks start at word 24");
    if (hipMemsetAsync(b->d_counters, 0, 24 * 8 + 256, b->stream) != hipSuccess) return fail(BSX_ERR_DEVICE);
    if (hipMalloc((void **)&b->d_scan_stats, 64 * 64) != hipSuccess) return fail(BSX_ERR_NOMEM);
    if (hipMemsetAsync(b->d_scan_stats, 0, 64 * 64, b->stream) != hipSuccess) return fail(BSX_ERR_DEVICE);
    if ((rc = ensure_scratch(b)) != BSX_OK) return fail(rc);
    b->counted = true;
    *out = b;
    return BSX_OK;
}

extern "C" void bsx_batch_destroy(bsx_batch *b)
{
    if (!b) return;
    (void)hipSetDevice(b->ref->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    if (b->sig_hist) bsx_sig_hist_report();
    if (b->sector_stats) bsx_sector_report();
    for (int m = 0; m < 2; m++)
        for (void *q : {(void *)b->d_seq[m], (void *)b->d_qual[m], (void *)b->d_off[m], (void *)b->d_cc[m]})
            if (q) (void)hipFree(q);
    for (void *q : {(void *)b->d_hits, (void *)b->d_pairs, (void *)b->d_npairs, (void *)b->d_scratch, (void *)b->d_dbg, (void *)b->d_queue, (void *)b->d_counters, (void *)b->d_scan_stats, (void *)b->d_cycles, (void *)b->d_heavy_list, (void *)b->d_heavy_count,
                    (void *)b->d_hstate, (void *)b->d_hslabs, (void *)b->d_htasks, (void *)b->d_htout, (void *)b->d_hactive[0], (void *)b->d_hactive[1], (void *)b->d_hcnt, (void *)b->d_redo})
        if (q) (void)hipFree(q);
    for (int m = 0; m < 2; m++) for (void *q : {(void *)b->d_hist_seq[m], (void *)b->d_hist_qual[m], (void *)b->d_hist_off[m]}) if (q) (void)hipFree(q);
    for (void *q : {(void *)b->d_leak_rec, (void *)b->d_leak_meta[0], (void *)b->d_leak_meta[1], (void *)b->d_leak_blk, (void *)b->d_leak_init, (void *)b->d_leak_final}) if (q) (void)hipFree(q);
    if (b->h_pinned) (void)hipHostFree(b->h_pinned);
    if (b->ev0) (void)hipEventDestroy(b->ev0);
    if (b->ev1) (void)hipEventDestroy(b->ev1);
    for (int g = 0; g < BSX_MAX_GROUPS; g++) {
        bsx_batch::Group &q = b->grp[g];
        if (q.s_ctrl) { (void)hipStreamSynchronize(q.s_ctrl); (void)hipStreamDestroy(q.s_ctrl); }
        for (hipEvent_t e : {q.ev_ctrl, q.ev_scan, q.ev_poll[0], q.ev_poll[1], q.ev_poll[2], q.ev_poll[3]}) if (e) (void)hipEventDestroy(e);
        for (void *p_ : {(void *)q.d_bins, (void *)q.d_bstart, (void *)q.d_chunk_tot, (void *)q.d_rank, (void *)q.d_order, (void *)q.d_glist}) if (p_) (void)hipFree(p_);
    }
    if (b->ev_sync) (void)hipEventDestroy(b->ev_sync);
    if (b->ev_wait) (void)hipEventDestroy(b->ev_wait);
    for (hipEvent_t e : b->scan_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : b->ctrl_ev) (void)hipEventDestroy(e);
    if (b->ev_align) (void)hipEventDestroy(b->ev_align);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    if (b->counted && b->ref->n_batches > 0) b->ref->n_batches--;
    delete b;
}

static int upload_mate(bsx_batch *b, int m, uint32_t n, const char *seqs, const uint64_t *off, const char *quals)
{
    const uint64_t bytes = off[n] - off[0];
    if (off[0] != 0) { g_bsx_err = "off[0] must be 0"; return BSX_ERR_ARG; }
    if (bytes + 256 > b->seq_cap[m]) {
        (void)hipFree(b->d_seq[m]); (void)hipFree(b->d_qual[m]);
        b->d_seq[m] = b->d_qual[m] = nullptr;
        b->seq_cap[m] = bytes + 256;
        HIP_TRY(hipMalloc((void **)&b->d_seq[m], b->seq_cap[m]));
        HIP_TRY(hipMalloc((void **)&b->d_qual[m], b->seq_cap[m]));
    }
    HIP_TRY(hipMemcpyAsync(b->d_seq[m], seqs, bytes, hipMemcpyHostToDevice, b->stream));
    if (quals) HIP_TRY(hipMemcpyAsync(b->d_qual[m], quals, bytes, hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(b->d_off[m], off, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, b->stream));
    return BSX_OK;
}

extern "C" int bsx_thread_device(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) { g_bsx_err = "device ordinal out of range"; return BSX_ERR_NODEVICE; }
    HIP_TRY(hipSetDevice(device));
    return BSX_OK;
}

extern "C" void *bsx_pinned_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) return nullptr;  // portable: every GPU of a -G list copies from / to it
    return p;
}
extern "C" void bsx_pinned_free(void *p) { if (p) (void)hipHostFree(p); }

extern "C" int bsx_batch_upload_se(bsx_batch *b, uint32_t n, const char *seqs, const uint64_t *off, const char *quals, uint32_t first_index)
{
    if (!b || !seqs || !off || b->paired || n > b->max_units) return BSX_ERR_ARG;
    HIP_TRY(hipSetDevice(b->ref->device));
    int rc = upload_mate(b, 0, n, seqs, off, quals);
    if (rc) return rc;
    b->n_units = n; b->first_index = first_index; b->has_qual = quals != nullptr; b->leak_meta_valid = false;
    HIP_TRY(stream_wait(b));  // host buffers may be reused by the caller
    return BSX_OK;
}

extern "C" int bsx_batch_upload_pe(bsx_batch *b, uint32_t n, const char *seqs_a, const uint64_t *off_a, const char *quals_a,
                                   const char *seqs_b, const uint64_t *off_b, const char *quals_b, uint32_t first_index)
{
    if (!b || !seqs_a || !off_a || !seqs_b || !off_b || !b->paired || n > b->max_units) return BSX_ERR_ARG;
    if ((quals_a == nullptr) != (quals_b == nullptr)) return BSX_ERR_ARG;
    HIP_TRY(hipSetDevice(b->ref->device));
    int rc = upload_mate(b, 0, n, seqs_a, off_a, quals_a);
    if (rc == BSX_OK) rc = upload_mate(b, 1, n, seqs_b, off_b, quals_b);
    if (rc) return rc;
    b->n_units = n; b->first_index = first_index; b->has_qual = quals_a != nullptr; b->leak_meta_valid = false;
    HIP_TRY(stream_wait(b));
    return BSX_OK;
}

extern "C" int bsx_batch_set_work_counters(bsx_batch *b, int on)
{
    if (!b) return BSX_ERR_ARG;
    b->work_counters = on ? 1 : 0;
    return BSX_OK;
}

extern "C" int bsx_batch_set_leak_exact(bsx_batch *b, int on)
{
    if (!b) return BSX_ERR_ARG;
    b->leak_exact = on ? 1 : 0;
    return BSX_OK;
}

extern "C" int bsx_batch_set_history(bsx_batch *b, uint32_t n, const char *seqs_a, const uint64_t *off_a, const char *quals_a, const char *seqs_b,
                                     const uint64_t *off_b, const char *quals_b)
{
    if (!b || n > 65536) return BSX_ERR_ARG;
    if (n && (!seqs_a || !off_a || (b->paired && (!seqs_b || !off_b)))) return BSX_ERR_ARG;
    HIP_TRY(hipSetDevice(b->ref->device));
    HIP_TRY(stream_wait(b));
    const char *sq[2] = {seqs_a, seqs_b}, *ql[2] = {quals_a, quals_b};
    const uint64_t *of[2] = {off_a, off_b};
    for (int m = 0; m < (b->paired ? 2 : 1); m++) {
        for (void *q : {(void *)b->d_hist_seq[m], (void *)b->d_hist_qual[m], (void *)b->d_hist_off[m]}) if (q) (void)hipFree(q);
        b->d_hist_seq[m] = b->d_hist_qual[m] = nullptr; b->d_hist_off[m] = nullptr;
        if (!n) continue;
        if (of[m][0] != 0) { g_bsx_err = "off[0] must be 0"; return BSX_ERR_ARG; }
        const uint64_t bytes = of[m][n];
        HIP_TRY(hipMalloc((void **)&b->d_hist_seq[m], bytes + 256));
        HIP_TRY(hipMalloc((void **)&b->d_hist_off[m], ((size_t)n + 1) * 8));
        HIP_TRY(hipMemcpy(b->d_hist_seq[m], sq[m], bytes, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(b->d_hist_off[m], of[m], ((size_t)n + 1) * 8, hipMemcpyHostToDevice));
        if (ql[m]) { HIP_TRY(hipMalloc((void **)&b->d_hist_qual[m], bytes + 256)); HIP_TRY(hipMemcpy(b->d_hist_qual[m], ql[m], bytes, hipMemcpyHostToDevice)); }
    }
    b->n_hist = n; b->leak_meta_valid = false;
    return BSX_OK;
}

extern "C" int bsx_batch_pool_sizes(const bsx_batch *b, uint32_t *units_per_round, uint32_t *task_pool)
{
    if (!b) return BSX_ERR_ARG;
    if (units_per_round) *units_per_round = b->hcap;
    if (task_pool) *task_pool = b->task_cap;
    return BSX_OK;
}

extern "C" int bsx_batch_set_debug(bsx_batch *b, int keep)
{
    if (!b) return BSX_ERR_ARG;
    HIP_TRY(hipSetDevice(b->ref->device));
    b->debug = keep == 1 ? 1 : 0;
    if (keep == 2 && !b->d_cycles) HIP_TRY(hipMalloc((void **)&b->d_cycles, (size_t)b->max_units * 4));
    if (keep == 0 && b->d_cycles) { (void)hipFree(b->d_cycles); b->d_cycles = nullptr; }
    if (keep == 1 && !b->d_dbg) HIP_TRY(hipMalloc((void **)&b->d_dbg, (size_t)b->max_units * 128));
    return ensure_scratch(b);
}

// exact mode: device arrays of the pre-pass and the argument fields that describe the mate streams
static int leak_prepare(bsx_batch *b, AlignArgs &A)
{
    const uint32_t cap = b->max_units + 65536u, nblk = cap / bsx_leak_blk() + 2;
    if (!b->d_leak_meta[0]) {
        for (int m = 0; m < (b->paired ? 2 : 1); m++) HIP_TRY(hipMalloc((void **)&b->d_leak_meta[m], (size_t)cap * 2));
        HIP_TRY(hipMalloc((void **)&b->d_leak_blk, (size_t)nblk * 16));
        HIP_TRY(hipMalloc((void **)&b->d_leak_final, bsx_leakstate_bytes()));
        HIP_TRY(hipMalloc((void **)&b->d_leak_rec, (size_t)b->max_units * 2 * bsx_leakrec_bytes()));
        b->leak_nblk = nblk;
    }
    for (int m = 0; m < 2; m++) { A.leak_meta[m] = b->d_leak_meta[m]; A.leak_blkmax[m] = b->d_leak_blk + (size_t)m * nblk; A.leak_blkset[m] = b->d_leak_blk + (size_t)(2 + m) * nblk; }
    A.leak_init = b->leak_has_init ? b->d_leak_init : nullptr;
    A.leak_rec = b->d_leak_rec;
    A.n_units_all = b->n_units;
    A.n_hist = b->n_hist;
    if (!b->leak_meta_valid) HIP_TRY(hipMemsetAsync(b->d_leak_blk, 0, (size_t)nblk * 16, b->stream));
    return BSX_OK;
}

static void fill_stream_args(bsx_batch *b, AlignArgs &A)
{
    memset(&A, 0, sizeof(A));
    bsx_fill_devparams(b->ref, A.P);
    A.first_index = b->first_index;
    for (int m = 0; m < 2; m++) {
        A.seq[m] = b->d_seq[m]; A.off[m] = b->d_off[m]; A.qual[m] = b->has_qual ? b->d_qual[m] : nullptr;
        A.hist_seq[m] = b->d_hist_seq[m]; A.hist_off[m] = b->d_hist_off[m]; A.hist_qual[m] = b->d_hist_qual[m];
    }
}

extern "C" int bsx_batch_set_leak_state(bsx_batch *b, const void *state, size_t bytes)
{
    if (!b || (state && bytes != bsx_leakstate_bytes())) return BSX_ERR_ARG;
    HIP_TRY(hipSetDevice(b->ref->device));
    HIP_TRY(stream_wait(b));
    if (!state) { b->leak_has_init = false; return BSX_OK; }
    if (!b->d_leak_init) HIP_TRY(hipMalloc((void **)&b->d_leak_init, bsx_leakstate_bytes()));
    HIP_TRY(hipMemcpy(b->d_leak_init, state, bytes, hipMemcpyHostToDevice));
    b->leak_has_init = true;
    return BSX_OK;
}

extern "C" int bsx_batch_get_leak_state(bsx_batch *b, void *state, size_t bytes)
{
    if (!b || !state || bytes != bsx_leakstate_bytes()) return BSX_ERR_ARG;
    if (b->n_units == 0) return BSX_ERR_STATE;
    if (b->ref->P.rrbs) { memset(state, 0, bytes); return BSX_OK; }   // RRBS plans every read from offset 0 (align.cpp:456): no state
    HIP_TRY(hipSetDevice(b->ref->device));
    AlignArgs A;
    fill_stream_args(b, A);
    A.n_units = b->n_units; A.first_unit = 0;
    int rc = leak_prepare(b, A);
    if (rc) return rc;
    bsx_launch_leak(A, b->paired, b->n_cu, !b->leak_meta_valid, false, b->d_leak_final, b->stream);
    HIP_TRY(hipGetLastError());
    b->leak_meta_valid = true;
    HIP_TRY(hipMemcpyAsync(state, b->d_leak_final, bytes, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(stream_wait(b));
    return BSX_OK;
}

extern "C" int bsx_batch_run(bsx_batch *b) { return b ? bsx_batch_run_range(b, 0, b->n_units) : BSX_ERR_ARG; }

extern "C" int bsx_batch_run_range(bsx_batch *b, uint32_t first_unit, uint32_t n_units)
{
    if (!b) return BSX_ERR_ARG;
    if (b->n_units == 0) return BSX_ERR_STATE;
    if (n_units == 0 || (uint64_t)first_unit + n_units > b->n_units) return BSX_ERR_ARG;
    HIP_TRY(hipSetDevice(b->ref->device));
    AlignArgs A;
    memset(&A, 0, sizeof(A));
    bsx_fill_devparams(b->ref, A.P);
    A.n_units = first_unit + n_units; A.first_index = b->first_index; A.debug = b->debug; A.rowcap = b->rowcap; A.kcap = key_cap(b->ref->P, b->rowcap); A.hbits = hset_bits(b->ref->P);
    A.hkcap = b->hkcap; A.hhbits = BSX_HSET_BITS; A.hslab_bytes = b->hslab_bytes;
    A.redo_count = b->d_redo; A.redo_list = b->d_redo + 1; A.unit_list = nullptr;
    A.first_unit = first_unit;
    for (int m = 0; m < 2; m++) { A.seq[m] = b->d_seq[m]; A.off[m] = b->d_off[m]; A.qual[m] = b->has_qual ? b->d_qual[m] : nullptr; A.cc[m] = b->d_cc[m]; }
    A.hits_out = b->d_hits; A.pairs_out = b->d_pairs; A.npairs_out = b->d_npairs;
    A.scratch = b->d_scratch; A.slab_bytes = b->slab_bytes; A.queue = b->d_queue; A.counters = b->d_counters; A.scan_stats = b->d_scan_stats; A.work_counters = (uint32_t)b->work_counters; A.dbg_plan = b->d_dbg; A.dbg_cycles = b->d_cycles; A.dbg_cat = b->d_cycles ? b->d_counters + 24 : nullptr;
    A.heavy_list = b->d_heavy_list; A.heavy_count = b->d_heavy_count;
    A.leak_exact = b->leak_exact; A.n_hist = b->leak_exact ? b->n_hist : 0; A.n_units_all = b->n_units;
    for (int m = 0; m < 2; m++) { A.hist_seq[m] = b->d_hist_seq[m]; A.hist_off[m] = b->d_hist_off[m]; A.hist_qual[m] = b->d_hist_qual[m]; }
    A.heavy_threshold = heavy_threshold_for(b->ref->P);
    HIP_TRY(hipMemsetAsync(b->d_queue, 0, 4, b->stream));
    HIP_TRY(hipMemsetAsync(b->d_heavy_count, 0, 4, b->stream));
    HIP_TRY(hipMemsetAsync(b->d_redo, 0, 4, b->stream));
    if (b->debug) HIP_TRY(hipMemsetAsync(b->d_scratch, 0, (size_t)b->max_units * b->slab_bytes, b->stream));
    HIP_TRY(hipEventRecord(b->ev0, b->stream));
    if (b->leak_exact && !b->ref->P.rrbs) {  // pre-pass of the exact mode: planner state that leaks from earlier reads
        int rc = leak_prepare(b, A);
        if (rc) return rc;
        bsx_launch_leak(A, b->paired, b->n_cu, !b->leak_meta_valid, true, nullptr, b->stream);
        HIP_TRY(hipGetLastError());
        b->leak_meta_valid = true;
    }
    bsx_launch_align(A, b->paired, b->grid_blocks, b->stream);
    HIP_TRY(hipGetLastError());
    if (b->stage_timing) {
        if (!b->ev_align) HIP_TRY(hipEventCreate(&b->ev_align));
        HIP_TRY(hipEventRecord(b->ev_align, b->stream));
    }
    b->last_heavy = 0; b->last_heavy_iters = 0; b->last_redo = 0; b->scan_ev_used = 0; b->ctrl_ev_used = 0;
    if (A.heavy_threshold) {
        // heavy pipeline: passes of k_hctrl / k_hscan until every deferred unit is finished (this call returns once the last pass
        // is queued and known to be the last; units that were not deferred are already complete)
        HIP_TRY(hipMemcpyAsync(b->h_pinned, b->d_heavy_count, 4, hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(stream_wait(b));
        const uint32_t n_heavy = b->h_pinned[0];
        b->last_heavy = n_heavy;
        // Each round handles up to hcap deferred units, split into unit groups whose passes run out of phase.  One pass of a group =
        //   k_hctrl (advance every active unit of the group, publish scan tasks)          on the group's high-priority stream
        //   task order (bsx_launch_task_order) + k_hscan over the published tasks          on the batch's stream
        // chained by events on the device: the counts a control pass leaves (active units, tasks) stay there — the next control
        // pass reads its input count from memory, the scan launches are sized for the whole task pool and their surplus blocks exit.
        // The host enqueues `chunk_passes` passes per group at a time, keeps two such chunks queued ahead, and only looks at a
        // group's active count once per chunk (polling an event, asleep in between) to learn when the group is finished; passes that were queued
        // beyond that point find nothing to do.  While k_hscan evaluates the tasks of one group, the control kernels of the others
        // run beside it — they are latency-bound chains of a few thousand waves.
        const bool shared_scan = b->ref->P.rrbs != 0;  // RRBS: runs of tasks over one window, scanned together
        // WGBS: the tasks of one window AND read offset share fetch and shift (k_hscan_same); BSX_SAME=0: one task per wave (k_hscan)
        // RRBS lists go through the same kernel with BSX_SAME=2 (scan 75.9 against 90.5 ms per step; the step does not move — 142.2 against 141.0 ms —, it
        // waits for the control passes there: k_hscan_shared stays the default for RRBS)
        const bool same_scan = b->same_kernel;   // (decided when the batch was created: its group list exists or not)
        const uint32_t spread = getenv("BSX_SPREAD") ? (uint32_t)atoi(getenv("BSX_SPREAD")) : (same_scan ? 1u : 0u);
        const int n_groups = b->n_groups;
        struct Group { uint32_t n0 = 0, passes = 0, polls = 0, polled = 0; int cur = 0; bool done = true, tail = false; HeavyArgsRaw H; uint32_t *blk[2]; };
        volatile uint32_t *pinned = (volatile uint32_t *)b->h_pinned;
        // (rounds of equal size: the reads of one window and offset that a pass holds are what k_hscan_same groups — a full round followed by
        //  a short one would scan the short one's candidates in small groups)
        const uint32_t n_rounds = (n_heavy + b->hcap - 1) / b->hcap, per_round = n_rounds ? (n_heavy + n_rounds - 1) / n_rounds : 0;
        for (uint32_t base = 0; base < n_heavy; base += per_round) {
            const uint32_t n_round = std::min(per_round, n_heavy - base);
            Group G[BSX_MAX_GROUPS];
            HIP_TRY(hipEventRecord(b->ev_sync, b->stream));             // everything queued so far (k_align, earlier rounds)
            auto enqueue_pass = [&](int g) -> int {
                Group &q = G[g];
                bsx_batch::Group &hw = b->grp[g];
                uint32_t *in = q.blk[q.cur], *out = q.blk[q.cur ^ 1];
                const bool fresh = q.passes == 0;
                if (fresh) {
                    HIP_TRY(hipStreamWaitEvent(hw.s_ctrl, b->ev_sync, 0));
                    HIP_TRY(hipMemsetAsync(out, 0, (size_t)BSX_HCNT_BLOCK * 4, hw.s_ctrl));
                } else HIP_TRY(hipStreamWaitEvent(hw.s_ctrl, hw.ev_scan, 0));  // the scan of the previous pass (its first kernel also cleared `out`)
                q.H.active_in = b->d_hactive[q.cur] + q.H.hidx_base; q.H.active_out = b->d_hactive[q.cur ^ 1] + q.H.hidx_base;
                q.H.n_active_in_ptr = in; q.H.n_active_in = q.n0; q.H.n_active_out = out; q.H.n_tasks = out + BSX_HCNT_TASKS; q.H.queue = out + BSX_HCNT_QUEUE;   // (apart: see BSX_HCNT_*)
                q.H.fresh = fresh ? 1 : 0;
                while (b->stage_timing && b->ctrl_ev_used + 3 > b->ctrl_ev.size()) { hipEvent_t ev_new = nullptr; HIP_TRY(hipEventCreate(&ev_new)); b->ctrl_ev.push_back(ev_new); }
                if (b->stage_timing) HIP_TRY(hipEventRecord(b->ctrl_ev[b->ctrl_ev_used], hw.s_ctrl));
                // (no more blocks than are resident at once — two per CU by their LDS —: blocks of a high-priority kernel that wait for a slot
                //  keep the dispatcher from placing the kernels of the normal-priority stream, 300 us per pass when the grid was twice that)
                bsx_launch_hctrl(A, q.H, b->paired, (int)std::min<uint32_t>((q.n0 + 3) / 4, (uint32_t)b->n_cu * b->hctrl_blocks_per_cu), hw.s_ctrl);
                HIP_TRY(hipGetLastError());
                if (b->stage_timing) HIP_TRY(hipEventRecord(b->ctrl_ev[b->ctrl_ev_used + 1], hw.s_ctrl));
                // scan order of the tasks this pass published, still on the group's stream: done by the time the main stream gets to the scan
                q.H.order = hw.d_order; q.H.xcd_map = b->xcd_map; q.H.ghead = hw.d_rank; q.H.glist = hw.d_glist;
                bsx_launch_task_order(q.H, b->bin_shift, b->n_bins, hw.d_bins, hw.d_bstart, hw.d_chunk_tot, hw.d_rank, hw.d_order, in, hw.s_ctrl, spread, same_scan);
                // The scan: on the batch's stream with a grid for the whole task pool — or, once the group is in its tail (few tasks per
                // pass, see the poll loop), behind the control kernel on the group's own high-priority stream with a small grid whose
                // blocks sweep: beside ANOTHER batch's bulk scans a pool-sized grid of mostly empty blocks only trickles through the
                // dispatcher, which stretched the tail of the older batch until the younger one's bulk was done — two batches in flight
                // always finished together, and their transfers never overlapped the other's kernels.
                if (b->stage_timing) { HIP_TRY(hipEventRecord(b->ctrl_ev[b->ctrl_ev_used + 2], hw.s_ctrl)); b->ctrl_ev_used += 3; }
                if (b->sig_hist) bsx_sig_hist_pass(q.H, hw.s_ctrl);
                hipStream_t s_scan = b->stream;
                if (q.tail) s_scan = hw.s_ctrl;
                else {
                    HIP_TRY(hipEventRecord(hw.ev_ctrl, hw.s_ctrl));
                    HIP_TRY(hipStreamWaitEvent(b->stream, hw.ev_ctrl, 0));
                }
                if (b->scan_ev_used + 2 > b->scan_ev.size()) {
                    hipEvent_t e0, e1;
                    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
                    b->scan_ev.push_back(e0); b->scan_ev.push_back(e1);
                }
                HIP_TRY(hipEventRecord(b->scan_ev[b->scan_ev_used], s_scan));
                if (same_scan) bsx_launch_hscan_same(A, q.H, s_scan, q.tail ? b->tail_grid_tasks : 0);
                else if (shared_scan) bsx_launch_hscan_shared(A, q.H, s_scan, q.tail ? b->tail_grid_tasks : 0);
                else bsx_launch_hscan(A, q.H, s_scan, q.tail ? b->tail_grid_tasks : 0);
                HIP_TRY(hipGetLastError());
                if (b->sector_stats) bsx_sector_pass(b->ref, s_scan);
                HIP_TRY(hipEventRecord(b->scan_ev[b->scan_ev_used + 1], s_scan));
                b->scan_ev_used += 2;
                HIP_TRY(hipEventRecord(hw.ev_scan, s_scan));
                q.cur ^= 1; q.passes++;
                b->last_heavy_iters++;
                return BSX_OK;
            };
            // one chunk: `chunk_passes` passes of every open group, interleaved pass by pass — the main stream takes the scans in the order
            // they are queued, and the scan of group g's next pass can only start after that group's control pass, which runs beside
            // the scans of the other groups — then the counts each group's last control pass of the chunk left: {active units, tasks}
            auto enqueue_chunk = [&]() -> int {
                for (int k = 0; k < b->chunk_passes; k++)
                    for (int g = 0; g < n_groups; g++)
                        if (!G[g].done) { int rc = enqueue_pass(g); if (rc) return rc; }
                for (int g = 0; g < n_groups; g++) {
                    Group &q = G[g];
                    if (q.done) continue;
                    const uint32_t slot = q.polls % BSX_POLL_SLOTS;
                    HIP_TRY(hipMemcpyAsync(b->h_pinned + 32 + 16 * g + 2 * slot, q.blk[q.cur], 4, hipMemcpyDeviceToHost, b->grp[g].s_ctrl));
                    HIP_TRY(hipMemcpyAsync(b->h_pinned + 32 + 16 * g + 2 * slot + 1, q.blk[q.cur] + BSX_HCNT_TASKS, 4, hipMemcpyDeviceToHost, b->grp[g].s_ctrl));
                    HIP_TRY(hipEventRecord(b->grp[g].ev_poll[slot], b->grp[g].s_ctrl));
                    q.polls++;
                }
                return BSX_OK;
            };
            int n_open = 0;
            for (int g = 0; g < n_groups; g++) {
                Group &q = G[g];
                const uint32_t lo = (uint32_t)((uint64_t)n_round * g / n_groups), hi = (uint32_t)((uint64_t)n_round * (g + 1) / n_groups);
                q.n0 = hi - lo; q.done = q.n0 == 0;
                memset(&q.H, 0, sizeof(q.H));
                const uint32_t tcap = b->task_cap / n_groups, toff = tcap * g;
                q.H.state = b->d_hstate; q.H.slabs = b->d_hslabs;
                q.H.tasks = b->d_htasks + (size_t)toff * bsx_htask_bytes(); q.H.tout = b->d_htout + (size_t)toff * bsx_htaskout_bytes();
                q.H.task_cap = tcap; q.H.list_base = base; q.H.hidx_base = lo;
                q.blk[0] = b->d_hcnt + (size_t)(2 * g) * BSX_HCNT_BLOCK; q.blk[1] = b->d_hcnt + (size_t)(2 * g + 1) * BSX_HCNT_BLOCK;
                if (!q.done) n_open++;
            }
            for (int depth = 0; depth < 2; depth++) { int rc = enqueue_chunk(); if (rc) return rc; }
            while (n_open > 0) {
                for (int g = 0; g < n_groups; g++) {  // the oldest chunk in flight
                    Group &q = G[g];
                    if (q.done) continue;
                    const uint32_t slot = q.polled % BSX_POLL_SLOTS;
                    HIP_TRY(wait_event(b->grp[g].ev_poll[slot]));
                    const uint32_t n_act = pinned[32 + 16 * g + 2 * slot], n_tasks = pinned[32 + 16 * g + 2 * slot + 1];
                    q.polled++;
                    if (b->trace) fprintf(stderr, "[bsx heavy] paired %d base %u group %d passes %u active %u tasks %u\n", b->paired, base, g, q.polled * b->chunk_passes, n_act, n_tasks);
                    if (n_act == 0) { q.done = true; n_open--; }  // (passes already queued for this group find no active unit)
                    else {
                        // (tested before the tail decision: a stuck unit publishes few tasks per pass — exactly what the tail mode looks like)
                        if (q.passes > 200000) { g_bsx_err = "heavy pipeline did not converge"; return BSX_ERR_DEVICE; }
                        if (b->tail_tasks && n_tasks < b->tail_tasks) q.tail = true;  // (the counts are two chunks old: a later pass may publish more — its blocks then sweep)
                    }
                }
                if (n_open > 0) { int rc = enqueue_chunk(); if (rc) return rc; }
            }
            for (int g = 0; g < n_groups; g++) {   // the main stream continues behind the last control pass of every group
                if (G[g].passes == 0) continue;
                HIP_TRY(hipEventRecord(b->grp[g].ev_ctrl, b->grp[g].s_ctrl));
                HIP_TRY(hipStreamWaitEvent(b->stream, b->grp[g].ev_ctrl, 0));
            }
        }
    }
    if (A.heavy_threshold && b->last_heavy) {
        // deferred units whose small duplicate set overflowed (single-end RRBS only, see k_hctrl): the main kernel redoes them,
        // undeferred, with its large per-wave set
        HIP_TRY(hipMemcpyAsync(b->h_pinned, b->d_redo, 4, hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(stream_wait(b));
        const uint32_t n_redo = b->h_pinned[0];
        b->last_redo = n_redo;
        if (n_redo) {
            AlignArgs R = A;
            R.unit_list = b->d_redo + 1; R.first_unit = 0; R.n_units = n_redo; R.heavy_threshold = 0;
            HIP_TRY(hipMemsetAsync(b->d_queue, 0, 4, b->stream));
            bsx_launch_align(R, b->paired, std::min<int>(b->grid_blocks, (int)((n_redo + 3) / 4)), b->stream);
            HIP_TRY(hipGetLastError());
        }
    }
    HIP_TRY(hipEventRecord(b->ev1, b->stream));
    b->ran = true;
    return BSX_OK;
}

extern "C" int bsx_batch_sync(bsx_batch *b)
{
    if (!b) return BSX_ERR_ARG;
    HIP_TRY(hipSetDevice(b->ref->device));
    HIP_TRY(stream_wait(b));
    return BSX_OK;
}

extern "C" float bsx_batch_kernel_ms(bsx_batch *b)
{
    if (!b || !b->ran) return -1.f;
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, b->ev0, b->ev1) != hipSuccess) return -1.f;
    return ms;
}

extern "C" int bsx_batch_scan_ms(bsx_batch *b, float *total_ms, uint32_t *launches)
{
    if (!b || !total_ms || !launches) return BSX_ERR_ARG;
    if (!b->ran) return BSX_ERR_STATE;
    HIP_TRY(hipSetDevice(b->ref->device));
    HIP_TRY(stream_wait(b));
    float sum = 0.f;
    for (size_t i = 0; i + 1 < b->scan_ev_used; i += 2) { float ms = 0.f; HIP_TRY(hipEventElapsedTime(&ms, b->scan_ev[i], b->scan_ev[i + 1])); sum += ms; }
    *total_ms = sum; *launches = (uint32_t)(b->scan_ev_used / 2);
    return BSX_OK;
}

extern "C" int bsx_batch_set_stage_timing(bsx_batch *b, int on)
{
    if (!b) return BSX_ERR_ARG;
    b->stage_timing = on != 0;
    return BSX_OK;
}
extern "C" int bsx_batch_stage_ms(bsx_batch *b, float out4[4], uint32_t *control_passes)
{
    if (!b || !out4) return BSX_ERR_ARG;
    if (!b->ran || !b->stage_timing || !b->ev_align) return BSX_ERR_STATE;
    HIP_TRY(hipSetDevice(b->ref->device));
    HIP_TRY(stream_wait(b));
    out4[0] = out4[1] = out4[2] = out4[3] = 0.f;
    if (b->ev_align) HIP_TRY(hipEventElapsedTime(&out4[0], b->ev0, b->ev_align));
    for (size_t i = 0; i + 2 < b->ctrl_ev_used; i += 3) {   // (the batch's stream waits for every group's last control pass before ev1: all recorded)
        float a = 0.f, o = 0.f;
        HIP_TRY(hipEventElapsedTime(&a, b->ctrl_ev[i], b->ctrl_ev[i + 1]));
        HIP_TRY(hipEventElapsedTime(&o, b->ctrl_ev[i + 1], b->ctrl_ev[i + 2]));
        out4[1] += a; out4[2] += o;
    }
    for (size_t i = 0; i + 1 < b->scan_ev_used; i += 2) { float ms = 0.f; HIP_TRY(hipEventElapsedTime(&ms, b->scan_ev[i], b->scan_ev[i + 1])); out4[3] += ms; }
    if (control_passes) *control_passes = (uint32_t)(b->ctrl_ev_used / 3);
    return BSX_OK;
}

extern "C" int bsx_batch_results_se(bsx_batch *b, bsx_hit *out, bsx_class_counts *counts)
{
    if (!b || b->paired || !out) return BSX_ERR_ARG;
    if (!b->ran) return BSX_ERR_STATE;
    HIP_TRY(hipSetDevice(b->ref->device));
    HIP_TRY(stream_wait(b));
    HIP_TRY(hipMemcpy(out, b->d_hits, (size_t)b->n_units * sizeof(bsx_hit), hipMemcpyDeviceToHost));
    if (counts) HIP_TRY(hipMemcpy(counts, b->d_cc[0], (size_t)b->n_units * sizeof(bsx_class_counts), hipMemcpyDeviceToHost));
    return BSX_OK;
}

extern "C" int bsx_batch_results_pe(bsx_batch *b, bsx_pair *out, bsx_class_counts *ca, bsx_class_counts *cb, uint16_t *n_pairs31)
{
    if (!b || !b->paired || !out) return BSX_ERR_ARG;
    if (!b->ran) return BSX_ERR_STATE;
    HIP_TRY(hipSetDevice(b->ref->device));
    HIP_TRY(stream_wait(b));
    HIP_TRY(hipMemcpy(out, b->d_pairs, (size_t)b->n_units * sizeof(bsx_pair), hipMemcpyDeviceToHost));
    if (ca) HIP_TRY(hipMemcpy(ca, b->d_cc[0], (size_t)b->n_units * sizeof(bsx_class_counts), hipMemcpyDeviceToHost));
    if (cb) HIP_TRY(hipMemcpy(cb, b->d_cc[1], (size_t)b->n_units * sizeof(bsx_class_counts), hipMemcpyDeviceToHost));
    if (n_pairs31) {
        std::vector<uint16_t> tmp((size_t)b->n_units * 32);
        HIP_TRY(hipMemcpy(tmp.data(), b->d_npairs, tmp.size() * 2, hipMemcpyDeviceToHost));
        for (uint32_t u = 0; u < b->n_units; u++) memcpy(n_pairs31 + (size_t)u * 31, tmp.data() + (size_t)u * 32, 62);
    }
    return BSX_OK;
}

extern "C" int bsx_batch_counters(bsx_batch *b, uint64_t c[BSX_N_COUNTERS])
{
    if (!b || !c) return BSX_ERR_ARG;
    HIP_TRY(hipSetDevice(b->ref->device));
    HIP_TRY(stream_wait(b));
    HIP_TRY(hipMemcpy(c, b->d_counters, BSX_N_COUNTERS * 8, hipMemcpyDeviceToHost));
    uint64_t sh[64 * 8];
    HIP_TRY(hipMemcpy(sh, b->d_scan_stats, sizeof(sh), hipMemcpyDeviceToHost));
    uint64_t dg[3] = {0, 0, 0};
    for (int i = 0; i < 64; i++) { for (int k = 0; k < 4; k++) c[7 + k] += sh[i * 8 + k]; c[15] += sh[i * 8 + 4]; for (int k = 0; k < 3; k++) dg[k] += sh[i * 8 + 5 + k]; }
    if (b->sig_hist && c[15])
        fprintf(stderr, "[sighist] scan kernel: %.4f of the candidates in groups, mean group %.2f reads, in groups of >= 4 %.4f; evaluations of a read whose words and threshold equal an earlier member's of its group: %.4f of all\n", (double)c[15] / (double)std::max<uint64_t>(1, c[7]),
                (double)dg[0] / (double)c[15], (double)dg[1] / (double)c[7], (double)dg[2] / (double)c[7]);
    return BSX_OK;
}
extern "C" int bsx_batch_reset_counters(bsx_batch *b)
{
    if (!b) return BSX_ERR_ARG;
    HIP_TRY(hipSetDevice(b->ref->device));
    HIP_TRY(hipMemsetAsync(b->d_counters, 0, BSX_N_COUNTERS * 8, b->stream));
    HIP_TRY(hipMemsetAsync(b->d_scan_stats, 0, 64 * 64, b->stream));
    HIP_TRY(stream_wait(b));
    return BSX_OK;
}

extern "C" int bsx_batch_download_reads(bsx_batch *b, int mate, char *seqs, uint64_t *off)
{
    if (!b || mate < 0 || mate > (b->paired ? 1 : 0) || !off) return BSX_ERR_ARG;
    HIP_TRY(hipSetDevice(b->ref->device));
    HIP_TRY(stream_wait(b));
    HIP_TRY(hipMemcpy(off, b->d_off[mate], ((size_t)b->n_units + 1) * 8, hipMemcpyDeviceToHost));
    if (seqs) HIP_TRY(hipMemcpy(seqs, b->d_seq[mate], off[b->n_units], hipMemcpyDeviceToHost));
    return BSX_OK;
}

// ---- debug readers (valid only after a run with bsx_batch_set_debug(b, 1)) -------------------------------------
static uint8_t *unit_slab(bsx_batch *b, uint32_t unit) { return b->d_scratch + (size_t)unit * b->slab_bytes; }

extern "C" int bsx_batch_debug_hits(bsx_batch *b, uint32_t unit, int mate, int orient, int w, uint32_t *chr_loc, uint32_t cap)
{
    if (!b || !b->debug || unit >= b->n_units || !b->ran) return BSX_ERR_STATE;
    HIP_TRY(hipSetDevice(b->ref->device));
    HIP_TRY(stream_wait(b));
    const uint64_t nclass = (uint64_t)b->ref->P.max_snp_num + 1, rowcap = b->rowcap;
    const uint64_t mbytes = mate_bytes(b->ref->P, b->rowcap);
    bsx_class_counts cc;
    HIP_TRY(hipMemcpy(&cc, b->d_cc[mate] + unit, sizeof(cc), hipMemcpyDeviceToHost));
    uint32_t n = orient ? cc.n_chit[w] : cc.n_hit[w];
    if (n > cap) n = cap;
    std::vector<uint64_t> tmp(n);
    const uint8_t *src = unit_slab(b, unit) + (mate ? mbytes : 0) + ((uint64_t)(orient * (nclass + 1) + w) * rowcap) * 8;
    if (n) HIP_TRY(hipMemcpy(tmp.data(), src, (size_t)n * 8, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) { chr_loc[2 * i] = (uint32_t)(tmp[i] >> 32); chr_loc[2 * i + 1] = (uint32_t)tmp[i]; }
    return (int)n;
}

extern "C" int bsx_batch_debug_pairs(bsx_batch *b, uint32_t unit, int w, uint32_t *pairhits6, uint32_t cap)
{
    if (!b || !b->debug || !b->paired || unit >= b->n_units || !b->ran) return BSX_ERR_STATE;
    HIP_TRY(hipSetDevice(b->ref->device));
    HIP_TRY(stream_wait(b));
    const uint64_t rowcap = b->rowcap;
    const uint64_t mbytes = mate_bytes(b->ref->P, b->rowcap);
    uint16_t np[32];
    HIP_TRY(hipMemcpy(np, b->d_npairs + (size_t)unit * 32, 64, hipMemcpyDeviceToHost));
    uint32_t n = np[w];
    if (n > cap) n = cap;
    const uint8_t *src = unit_slab(b, unit) + 2 * mbytes + (uint64_t)w * rowcap * 24;
    if (n) HIP_TRY(hipMemcpy(pairhits6, src, (size_t)n * 24, hipMemcpyDeviceToHost));
    return (int)n;
}

extern "C" int bsx_batch_last_heavy_units(bsx_batch *b)
{
    if (!b || !b->ran) return BSX_ERR_STATE;
    HIP_TRY(hipSetDevice(b->ref->device));
    HIP_TRY(stream_wait(b));
    return (int)b->last_heavy;
}

extern "C" int bsx_batch_last_heavy_list(bsx_batch *b, uint32_t *units, uint32_t cap)
{
    if (!b || (!units && cap)) return BSX_ERR_ARG;
    if (!b->ran) return BSX_ERR_STATE;
    HIP_TRY(hipSetDevice(b->ref->device));
    HIP_TRY(stream_wait(b));
    const uint32_t n = std::min<uint32_t>(cap, (uint32_t)b->last_heavy);
    if (n) HIP_TRY(hipMemcpy(units, b->d_heavy_list, (size_t)n * 4, hipMemcpyDeviceToHost));
    return (int)n;
}

extern "C" int bsx_batch_last_redo_units(bsx_batch *b)
{
    if (!b || !b->ran) return BSX_ERR_STATE;
    HIP_TRY(hipSetDevice(b->ref->device));
    HIP_TRY(stream_wait(b));
    return (int)b->last_redo;
}

extern "C" int bsx_batch_ctrl_clocks(bsx_batch *b, uint64_t out[24])
{
    if (!b || !out) return BSX_ERR_ARG;
    HIP_TRY(hipSetDevice(b->ref->device));
    HIP_TRY(stream_wait(b));
    HIP_TRY(hipMemcpy(out, b->d_counters + 24, 192, hipMemcpyDeviceToHost));   // (behind the BSX_N_COUNTERS counters: A.dbg_cat)
    return BSX_OK;
}

extern "C" int bsx_batch_unit_cycles(bsx_batch *b, uint32_t *out)
{
    if (!b || !b->d_cycles || !b->ran || !out) return BSX_ERR_STATE;
    HIP_TRY(hipSetDevice(b->ref->device));
    HIP_TRY(stream_wait(b));
    HIP_TRY(hipMemcpy(out, b->d_cycles, (size_t)b->n_units * 4, hipMemcpyDeviceToHost));
    return BSX_OK;
}

extern "C" int bsx_batch_debug_plan(bsx_batch *b, uint32_t unit, int mate, int32_t *start32, int32_t *order32)
{
    if (!b || !b->debug || unit >= b->n_units || !b->ran) return BSX_ERR_STATE;
    HIP_TRY(hipSetDevice(b->ref->device));
    HIP_TRY(stream_wait(b));
    uint8_t buf[64];
    HIP_TRY(hipMemcpy(buf, b->d_dbg + (size_t)unit * 128 + (mate ? 64 : 0), 64, hipMemcpyDeviceToHost));
    for (int i = 0; i < 32; i++) { start32[i] = buf[i]; order32[i] = buf[32 + i]; }
    return BSX_OK;
}

extern "C" int bsx_batch_synth_reads_kind(bsx_batch *b, uint32_t n, uint32_t read_len, uint64_t seed, uint32_t first_index, int kind)
{
    if (!b || n == 0 || n > b->max_units || read_len < 16 || read_len > 160 || kind < 0 || kind > 2) return BSX_ERR_ARG;
    HIP_TRY(hipSetDevice(b->ref->device));
    const int nm = b->paired ? 2 : 1;
    std::vector<uint64_t> off((size_t)n + 1);
    for (uint32_t i = 0; i <= n; i++) off[i] = (uint64_t)i * read_len;
    for (int m = 0; m < nm; m++) {
        if (off[n] + 256 > b->seq_cap[m]) return BSX_ERR_ARG;
        HIP_TRY(hipMemcpy(b->d_off[m], off.data(), off.size() * 8, hipMemcpyHostToDevice));
    }
    const bool q = kind == 1;  // the trimming workload carries qualities (low-quality 3' tails)
    int rc = bsx_synth_reads_launch(b->ref, n, read_len, b->paired, seed, first_index, b->d_seq[0], b->d_seq[1], b->stream, kind, q ? b->d_qual[0] : nullptr,
                                    q && b->paired ? b->d_qual[1] : nullptr);
    if (rc) return rc;
    b->n_units = n; b->first_index = first_index; b->has_qual = q ? 1 : 0; b->leak_meta_valid = false;
    return BSX_OK;
}

extern "C" int bsx_batch_synth_reads(bsx_batch *b, uint32_t n, uint32_t read_len, uint64_t seed, uint32_t first_index)
{
    return bsx_batch_synth_reads_kind(b, n, read_len, seed, first_index, 0);
}

extern "C" int bsx_batch_download_quals(bsx_batch *b, int mate, char *quals)
{
    if (!b || mate < 0 || mate > (b->paired ? 1 : 0) || !quals) return BSX_ERR_ARG;
    if (!b->has_qual) return BSX_ERR_STATE;
    HIP_TRY(hipSetDevice(b->ref->device));
    HIP_TRY(stream_wait(b));
    uint64_t end = 0;
    HIP_TRY(hipMemcpy(&end, b->d_off[mate] + b->n_units, 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(quals, b->d_qual[mate], end, hipMemcpyDeviceToHost));
    return BSX_OK;
}
