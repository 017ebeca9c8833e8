// bsx_align.hip — the alignment hot path as ONE persistent gfx950 kernel.
//
// Reference call tree replaced (BSMAP v2.6 file:line):
//   SingleAlign::Do_Batch align.cpp:591 / PairAlign::Do_Batch pairs.cpp:192
//     FilterReads align.cpp:579  (TrimAdapter :371, TrimLowQual :59, CountNs :48)
//     RunAlign align.cpp:435 / PairAlign::RunAlign pairs.cpp:137
//       ConvertBinaySeq align.cpp:90, ReorderSeed align.cpp:454 (+ AdjustSeedStartArray :506, CountSeeds :549)
//       SnpAlign align.cpp:168 (+ CountMismatch align.h:167, RefSeq::int2hit dbseq.cpp:585)
//       SortHits4PE align.cpp:363, GetPairs pairs.cpp:34
//     StringAlign align.cpp:610 / StringAlignPair pairs.cpp:222 / StringAlignUnpair pairs.cpp:244 (selection only)
//
// Mapping onto CDNA4: one 64-lane wavefront owns one read (or read pair) at a time and pulls the next unit from a
// global queue (persistent grid: the reference's "thread grabs the next batch" loop, main.cpp:49-73, at wave
// granularity).  Inside a unit every data-parallel step is spread over the 64 lanes:
//   * seed hashes + bucket-header gathers: lane = read offset (one dependent HBM round trip for all ~130 offsets)
//   * candidate scan: the (phase x strand) bucket ranges of a round are concatenated into one virtual list that the
//     wave walks 64 candidates at a time; each lane loads its index entry (coalesced), the first 16 bytes of
//     reference at entry+h (random), funnel-shifts them to the read's frame and counts mismatches with
//     xor/and/popcount; only lanes that survive the first 48 nt load the rest
//   * the reference's order-dependent tail (duplicate suppression, -w cap, -r 0 early return) is replayed over the
//     surviving lanes in lane order (= list order) with wave-uniform state, so results are bit-identical
//   * PE: class lists are rank-sorted across lanes; the window join runs lane-parallel over the b-list
// Read words and masks live in SGPRs during the scan; planner tables live in LDS; hit lists live in a per-wave HBM
// slab (they are usually 1-2 entries but may legally reach 1000 per class).
#include "bsx_internal.h"
#include "bsx_dev.h"
#include "bsx_kernel_args.h"

namespace {

typedef unsigned long long u64;

struct __attribute__((packed, aligned(4))) U4 { uint32_t a, b, c, d; };
struct __attribute__((packed, aligned(4))) U2 { uint32_t a, b; };

// Loads of the main kernel (k_align) and the control kernel: random one-touch gathers over the 172 MB bucket table, the
// 5.9 GB of entries and the 1.5 GB reference.  -DBSX_MAIN_NT=1 marks them non-temporal so that they do not push the scan
// kernel's shared lines out of L2 while the two run side by side (two batches in flight).
#ifndef BSX_MAIN_NT
#define BSX_MAIN_NT 0
#endif
#ifndef BSX_MAIN_PREFETCH
#define BSX_MAIN_PREFETCH 1  /* the main kernel's scan requests the next chunk's index entries before this chunk's reference gather (wave_scan_range) */
#endif
typedef uint32_t v4u_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t v2u_a4 __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ U4 ldm4(const uint32_t *p)
{
#if BSX_MAIN_NT
    const v4u_a4 v = __builtin_nontemporal_load(reinterpret_cast<const v4u_a4 *>(p));
    return U4{v.x, v.y, v.z, v.w};
#else
    return *reinterpret_cast<const U4 *>(p);
#endif
}
__device__ __forceinline__ U2 ldm2(const uint32_t *p)
{
#if BSX_MAIN_NT
    const v2u_a4 v = __builtin_nontemporal_load(reinterpret_cast<const v2u_a4 *>(p));
    return U2{v.x, v.y};
#else
    return *reinterpret_cast<const U2 *>(p);
#endif
}
__device__ __forceinline__ uint32_t ldm1(const uint32_t *p)
{
#if BSX_MAIN_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

// wave64 ballot straight from the condition bit (HIP's __ballot(int) goes through a 0/1 VGPR and a compare)
__device__ __forceinline__ unsigned long long bsx_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// x * 30 + c on the low 24 bits of x in one full-rate instruction (a plain 32-bit multiply is quarter rate); only the
// low 5 bits of the result are used (shift amounts), and 30 = -2 mod 32
#define mad30(x, c) ({ uint32_t d_; asm("v_mad_u32_u24 %0, %1, 30, " #c : "=v"(d_) : "v"(x)); d_; })  /* x * 30 + c (low 24 bits of x) */
__device__ __forceinline__ uint32_t rfl(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint32_t rl(uint32_t x, int l) { return __builtin_amdgcn_readlane(x, l); }
__device__ __forceinline__ uint32_t rl_u(uint32_t x, uint32_t l) { return __builtin_amdgcn_readlane(x, (int)__builtin_amdgcn_readfirstlane(l)); }  // wave-uniform lane number held in a register
__device__ __forceinline__ u64 rl64(u64 x, int l) { return ((u64)rl((uint32_t)(x >> 32), l) << 32) | rl((uint32_t)x, l); }
__device__ __forceinline__ u64 lanemask_lt(int lane) { return lane ? (~0ull >> (64 - lane)) : 0ull; }
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

// the first lane that holds the minimum of v (0xffffffff: the lane does not compete), -1 if none competes: the minimum of (v << 8 | lane) over the wave
__device__ __forceinline__ int wave_argmin_u32(uint32_t v)
{
    // minimum of every row of 16 lanes by four shifted v_min_u32 (lane 15 of the row holds it), the four rows joined on the scalar unit, the lane by a ballot
    // (correct on the box: tools/experiments/r05/dpp_min_test.hip; the 64-bit shuffle reduction it replaces was 18 vector and 12 LDS instructions per call, ~40 calls per pair)
    uint32_t r = v;
#define BSX_DPP_MIN(ctrl) r = min(r, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)r, ctrl, 0xf, 0xf, false))
    BSX_DPP_MIN(0x111); BSX_DPP_MIN(0x112); BSX_DPP_MIN(0x114); BSX_DPP_MIN(0x118);   // row_shr 1, 2, 4, 8
#undef BSX_DPP_MIN
    const uint32_t m = min(min(rl(r, 15), rl(r, 31)), min(rl(r, 47), rl(r, 63)));
    return m == 0xffffffffu ? -1 : (int)__builtin_ctzll(bsx_ballot(v == m));
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// The wave-uniform part of a read's state lives in LDS (one MateU per mate and wave, allocated by the kernels): as members of a
// private struct these fields ended up in scratch memory — the 128-VGPR budget of the main kernel has no room for them — and
// every access was a scratch round trip inside a latency-bound chain.  All lanes of the wave write the same value.
struct MateU {
    int len, raw_len, max_snp, seedseg, filtered;
    int nfull;           // len / seed_size (RRBS: cmodeindex = nfull - 1 - segment, align.cpp:221)
    uint32_t flags;      // bit0 flag_chain, bit1 cflag_chain, bit2 duplicate-suppression set overflowed (RRBS single-end only)
    uint32_t snp_thres;
    uint32_t nkeys;
    uint32_t index;      // ReadInf.index
    int defer;           // main kernel: a candidate list exceeded heavy_threshold, redo this unit in the heavy kernel
    uint32_t pad;
};
struct MateLds {
    uint8_t seq[160];
    uint8_t qual[160];
    uint32_t w[2][10], m[2][10];  // packed read words / N-masks per orientation (align.h:74-77, copy 0 only)
    uint32_t cnt[2][160];         // index2[seed][0] for every read offset (CountSeeds' operand); [noff, noff+16): the stale tail (exact mode)
    uint8_t start[2][16];         // seed_start_array / cseed_start_array
    uint8_t order[2][16];         // seedindex[].second / cseedindex[].second
    // "-p 1 exact" mode: what the reference's never-reset planner state holds when this read is planned (resolve_leak)
    uint32_t stale_key[2][16];    // seed_array / cseed_array entries [noff, noff+16) left behind by earlier, longer reads
    uint8_t stale_so[2];          // seed_start_offset / cseed_start_offset of the last read that set them
    MateU u, u2;                  // the read's wave-uniform state (Mate::u points here); u2: the absent second mate of a single-end unit
    __attribute__((aligned(16))) uint32_t fl[4][8];   // context prefilter (wave_scan_range<.., CTX>): per index phase the read's 32 nt left and right of the seed and their T-masks
};

template <bool PE> struct WaveLds { MateLds mate[PE ? 2 : 1]; };

struct BlockLds {
    uint8_t prof[16][16];
    uint8_t nt_tab[256];   // per read byte: bits 0-1 the nt's code (alphabet), 2-3 the code of its complement (rev_alphabet), 4-5 reg_alphabet (3 for a nucleotide, 0 for N); N: codes of A / T
    uint32_t anchor[BSX_LDS_CHR + 1], chr_size[BSX_LDS_CHR], rc_offset[BSX_LDS_CHR];
};

// wave-uniform per-mate state; cnt_reg / key_reg are lane-distributed tables
typedef __attribute__((address_space(3))) MateU LdsMateU;
struct Mate {
    LdsMateU *u;
    uint32_t cnt_reg;    // lane (orient*16+w) holds _cur_n_hit / _cur_n_chit
    uint32_t key_reg;    // lane i holds the i-th accepted forward coordinate (hitset, first 64)
    uint32_t bloom0, bloom1;  // 4096-bit membership filter over all accepted coordinates (bit b of lane l)
};
__device__ __forceinline__ LdsMateU *lds_mate(MateU *p) { return (LdsMateU *)p; }

struct Slab {
    // Rows are laid out exactly like the reference's `new HitArray[MAXSNPS+1]` (align.cpp:21-22): row w starts at
    // w*(MAXHITS+1).  A class may legally exceed -w by one entry per extra SnpAlign call (PE levels, RRBS rounds); with
    // the default -w 1000 those entries land in the first slots of row w+1 — in the reference and, by construction, here.
    u64 *hits;        // [2][nclass+1][rowcap]
    uint32_t *keys;   // [(nclass+1)*rowcap] accepted coordinates beyond the 64 held in key_reg ...
    uint32_t *kslot;  // ... and the hash-set slot each of them occupies (for the per-unit clean-up)
    uint32_t *hset;   // [1 << hbits] open-addressing set of (coordinate+1); all zero between units
    u64 *tmp;         // [BSX_SORT_TMP] sort scratch
    uint32_t rowcap, nclass;
    // keys / kslot hold kcap entries.  WGBS: every remembered coordinate is also a hit, so the -w caps bound them ((nclass+1) rows
    // suffice).  RRBS single-end: coordinates the fragment-size filter rejects are remembered too (align.cpp:201-207: the
    // insert comes first) and no cap bounds those — such batches get a much larger set (bsx_api.hip) and a unit that still
    // overflows it is flagged BSX_F_LIMIT instead of writing past the slab
    uint32_t kcap, hbits;
    __device__ __forceinline__ u64 *list(int orient, int w) const { return hits + ((size_t)(orient * (nclass + 1) + w)) * rowcap; }
};

struct Counters { u64 n_lookup, n_cand, sum_w, n_orient; };

__device__ __forceinline__ int nt_idx(uint32_t c)
{
    c |= 0x20;
    return c == 'a' ? 0 : c == 'c' ? 1 : c == 'g' ? 2 : c == 't' ? 3 : -1;
}

__device__ __forceinline__ uint32_t n_of(const Mate &M, int orient, int w) { return rl(M.cnt_reg, orient * 16 + w); }

// ---------------------------------------------------------------------------------------------------------------
// FilterReads (align.cpp:579-589)
// ---------------------------------------------------------------------------------------------------------------
// stream position j of mate stream `mate`: unit j of the batch, or for j < 0 read n_hist + j of the attached history
__device__ void load_and_filter(const AlignArgs &A, MateLds &L, Mate &M, int mate, long j, int lane)
{
    const DevParams &P = A.P;
    const bool hist = j < 0;
    const uint64_t *off = hist ? A.hist_off[mate] : A.off[mate];
    const uint64_t jj = hist ? (uint64_t)((long)A.n_hist + j) : (uint64_t)j;
    const uint64_t b = off[jj], e = off[jj + 1];
    int len = (int)min((uint64_t)P.max_readlen, e - b);  // reads.cpp:115-117
    const uint8_t *sq = hist ? A.hist_seq[mate] : A.seq[mate], *qq = hist ? A.hist_qual[mate] : A.qual[mate];
    const uint8_t *s = sq + b, *q = qq ? qq + b : nullptr;
    for (int i = lane; i < 160; i += 64) {
        L.seq[i] = i < len ? s[i] : 0;
        L.qual[i] = (q && i < len) ? q[i] : 0;
    }
    wave_fence();
    M.u->raw_len = len;
    M.u->filtered = 0;
    // TrimAdapter, WGBS branch (align.cpp:410-423): first adapter, then first position, whose <=15-nt prefix matches
    // the read tail with k >= 5*mismatches and k > 3
    if (P.n_adapter > 0 && len >= 5) {
        bool done = false;
        for (int a = 0; a < P.n_adapter && !done; a++) {
            const int al = P.adapter_len[a];
            for (int base = P.seed_size; base < len - (P.rrbs ? 5 : 4) && !done; base += 64) {
                const int pos = base + lane;
                bool hit = false;
                if (pos < len - (P.rrbs ? 5 : 4)) {
                    int k = 0, m0 = 0;
                    for (; k < al && k < 15 && pos + k < len; k++)
                        if ((m0 += (P.adapter[a][k] != (char)L.seq[pos + k])) > 4) break;
                    if (!P.rrbs) hit = (k >= m0 * 5 && k > 3);
                    else if (k >= m0 * 5) {
                        // RRBS branch (align.cpp:375-408): the digestion-site remainder must precede the adapter
                        const int dl = P.digest_len;
                        int m = m0;
                        for (int t = 0; t < dl - P.digest_pos; t++) {
                            char an = P.digest_site[t], rn = (char)L.seq[pos - dl + P.digest_pos + t];
                            m += (an != rn) && (an != 'C' || rn != 'T');
                        }
                        hit = (k >= m * 5);
                        if (!hit && P.pairend) {
                            m = m0;
                            for (int t = 0; t < dl - P.digest_pos; t++) {
                                char an = P.digest_site[t], rn = (char)L.seq[pos - dl + P.digest_pos + t];
                                m += (an != rn) && (an != 'G' || rn != 'A');
                            }
                            hit = (k >= m * 5);
                        }
                    }
                }
                const u64 mask = bsx_ballot(hit);
                if (mask) { len = base + (int)__builtin_ctzll(mask); done = true; }
            }
        }
    }
    // TrimLowQual (align.cpp:59-79): cut after the last base whose quality exceeds zero_qual+threshold, if that keeps
    // >= seed_size bases; otherwise the read is rejected.  (The SAM rebasing at :64-67 shifts both sides equally.)
    const int qlen = q ? len : 0;
    if (P.qual_threshold != 0 && q && qlen != 1) {
        int best = 0;
        for (int base = 0; base < qlen; base += 64) {
            const int i = base + lane;
            const bool good = i < qlen && (int)(int8_t)L.qual[i] > P.zero_qual + P.qual_threshold;
            const u64 mask = bsx_ballot(good);
            if (mask) best = base + 64 - (int)__builtin_clzll(mask);
        }
        if (best >= P.seed_size) len = min(len, best);
        else M.u->filtered = 1;
    }
    if (len < P.seed_size) M.u->filtered = 1;  // min_read_size
    {
        uint32_t ns = 0;  // CountNs (align.cpp:48-55)
        for (int base = 0; base < len; base += 64) {
            const int i = base + lane;
            ns += (uint32_t)__builtin_popcountll(bsx_ballot(i < len && nt_idx(L.seq[i]) < 0));
        }
        if ((int)ns > P.max_ns) M.u->filtered = 1;
    }
    M.u->len = len;
    M.u->max_snp = M.u->filtered ? 0 : (int)(((uint64_t)(P.max_snp_num + 1) * (uint64_t)(len - 1)) / (uint64_t)M.u->raw_len);
    const int x = (len - P.index_interval + 1) / P.seed_size, y = M.u->max_snp + 1;  // align.cpp:440
    M.u->seedseg = M.u->filtered ? 0 : min(x, y);
    M.u->nfull = len / P.seed_size;
    M.u->snp_thres = (uint32_t)M.u->max_snp;
    M.u->nkeys = 0;
    M.cnt_reg = 0;
    M.key_reg = 0;
    M.bloom0 = M.bloom1 = 0;
    M.u->defer = 0;
    M.u->flags = 0;  // (set by pack_read; a filtered read never gets there and bit 2 is reported)
}

// ---------------------------------------------------------------------------------------------------------------
// ConvertBinaySeq (align.cpp:90-162): 2-bit words + N masks for the enabled orientations (no shifted copies:
// the scan shifts the reference instead)
// ---------------------------------------------------------------------------------------------------------------
#ifndef BSX_PACK_UNROLL
#define BSX_PACK_UNROLL 2   /* (unrolled eight times the sixteen table reads in flight tip the paired kernel's register allocation into its bad state: 51 spilled VGPRs against 35) */
#endif
__device__ void pack_read(const DevParams &P, const BlockLds &BL, MateLds &L, Mate &M, int readset, int lane, Counters &C)
{
    M.u->flags = ((P.chains || readset < 2) ? 1u : 0u) | ((P.chains || readset == 2) ? 2u : 0u);  // align.cpp:93-94
    // lane = orientation (1 bit), word (4 bits), half of the word (1 bit): eight nt per lane through the byte table, the halves joined across neighbouring lanes
    // (round 5: twenty lanes of sixteen nt each with the alphabet worked out per nt were 440 vector instructions per read of a kernel whose VALU is 0.7 busy)
    const int orient = lane >> 5, t = (lane >> 1) & 15, hf = lane & 1, len = M.u->len;
    uint32_t w = 0, m = 0;
    if (t < 10) {
        const int pos0 = t * 16 + hf * 8;
#pragma unroll BSX_PACK_UNROLL
        for (int j = 0; j < 8; j++) {
            const int pos = pos0 + j;
            uint32_t e = 0;
            if (pos < len) e = BL.nt_tab[L.seq[orient ? len - 1 - pos : pos]];
            w = (w << 2) | ((e >> (2 * orient)) & 3u);
            m = (m << 2) | (e >> 4);
        }
    }
    const uint32_t w_lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xf, 0xf, true), m_lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xf, 0xf, true);   // quad_perm [1, 0, 3, 2]: the neighbour's half
    if (t < 10 && hf == 0) {
        L.w[orient][t] = (w << 16) | w_lo;
        L.m[orient][t] = (m << 16) | m_lo;
    }
    wave_fence();
    C.n_orient += (M.u->flags & 1) + ((M.u->flags >> 1) & 1);
}

__device__ __forceinline__ uint32_t seed_key_at(const DevParams &P, const uint32_t *w, int o)
{
    const int q = o >> 4, r = o & 15;
    const u64 v = ((u64)w[q] << 32) | w[q + 1 < 10 ? q + 1 : 9];
    return bsx_seed_hash((uint32_t)(v >> (64 - 2 * P.seed_size - 2 * r)) & P.seed_bits);
}

// seed_array[o] / cseed_array[o] as the reference's planner sees it: the read's own hash for o < noff; beyond that the entry
// an earlier read left behind (only reachable in "-p 1 exact" mode, see resolve_leak; zero-state mode never reads past noff)
// (EXACT: the main kernel exists in two instantiations; with the mode off it carries none of this)
template <bool EXACT>
__device__ __forceinline__ uint32_t key_at(const DevParams &P, const MateLds &L, const Mate &M, int orient, int o)
{
    if (!EXACT) return seed_key_at(P, L.w[orient], o);
    const int noff = M.u->len - P.seed_size + 1;
    return o >= noff ? L.stale_key[orient][min(o - noff, 15)] : seed_key_at(P, L.w[orient], o);
}

// index2[s][0] (= 2 + bucket size, 0 for an empty bucket) for every offset of this orientation -> L.cnt[orient][]
template <bool EXACT>
__device__ void plan_counts(const DevParams &P, MateLds &L, const Mate &M, int orient, int lane, bool with_tail)
{
    const int noff = M.u->len - P.seed_size + 1, n = noff + (with_tail ? 16 : 0);
    for (int base = 0; base < n; base += 64) {
        const int o = base + lane;
        if (o < n) {
            const uint32_t key = key_at<EXACT>(P, L, M, orient, o);
            const U2 b = ldm2(P.bucket_off + key);
            const uint32_t c = b.b - b.a;
            L.cnt[orient][o] = P.rrbs ? c : (c ? c + 2 : 0);
        }
    }
    wave_fence();
}

// GetTotalSeedLoc of start offset (lane & 15) in every lane (align.cpp:458-468): the segments are dealt to the four groups of 16 lanes
// (there are fewer than 16 start offsets) and the partial sums added across the groups
__device__ __forceinline__ uint32_t start_totals(const DevParams &P, const BlockLds &BL, const uint32_t *cnt, int nseg, int nstart, int lane)
{
    const int I = P.index_interval, g = lane >> 4, sl = lane & 15;
    int tot = 0;
    if (sl < nstart)
        for (int seg = g; seg < nseg; seg += 4)
            for (int ph = 0; ph < I; ph++) tot += (int)cnt[BL.prof[seg][ph] + sl - ph];
    tot += __shfl_xor(tot, 16);
    tot += __shfl_xor(tot, 32);
    return lane < nstart ? (uint32_t)tot : 0xffffffffu;   // (only the lanes that ARE a start offset compete)
}

// GetTotalSeedLoc for every start offset, first minimum wins (align.cpp:458-468); counts must be in L.cnt
__device__ int plan_best_offset(const DevParams &P, const BlockLds &BL, const MateLds &L, const Mate &M, int orient, int lane)
{
    const int I = P.index_interval, nseg = M.u->seedseg, nstart = (M.u->len - I + 1) % P.seed_size;
    return wave_argmin_u32(start_totals(P, BL, L.cnt[orient], nseg, nstart, lane));
}

// ---------------------------------------------------------------------------------------------------------------
// ReorderSeed (align.cpp:454-504) for one orientation
// ---------------------------------------------------------------------------------------------------------------
template <bool EXACT>
__device__ void plan_orient(const DevParams &P, const BlockLds &BL, MateLds &L, const Mate &M, int orient, int lane, Counters &C, bool leaky_arg = false)
{
    const bool leaky_exact = EXACT && leaky_arg;
    const int I = P.index_interval, S = P.seed_size, nseg = M.u->seedseg;
    plan_counts<EXACT>(P, L, M, orient, lane, leaky_exact);
    const uint32_t *cnt = L.cnt[orient];
    int offset = leaky_exact ? (int)L.stale_so[orient] : 0;  // the loop below does not run for such a read: the old value stays (align.cpp:458)
    const int nstart = P.rrbs ? 0 : (M.u->len - I + 1) % S;
    u64 lookups = 0;
    if (nstart > 0) {  // GetTotalSeedLoc for every start, first minimum wins (align.cpp:458-468)
        const int best = wave_argmin_u32(start_totals(P, BL, cnt, nseg, nstart, lane));
        if (best >= 0) offset = best;
        lookups += (u64)nstart * nseg * I;
    }
    // AdjustSeedStartArray (align.cpp:506-528)
    if (lane < 16) L.start[orient][lane] = (uint8_t)offset;
    wave_fence();
    if (!P.rrbs) {
        for (int i = 0; i < nseg; i++) {
            const int ptr = (i % 2 == 0) ? i / 2 : nseg - 1 - i / 2;
            const int start = ptr == 0 ? 0 : L.start[orient][ptr - 1];
            const int end = ptr == nseg - 1 ? nstart : L.start[orient][ptr + 1];
            uint32_t tv = 0xffffffffu;
            const int ii = start + lane;
            if (ii <= end) {
                int tt = 0;
                for (int ph = 0; ph < I; ph++) tt += (int)cnt[BL.prof[ptr][ph] + ii - ph];
                tv = (uint32_t)tt;
            }
            const int best = wave_argmin_u32(tv);
            const int pick = best >= 0 ? start + best : start;
            if (lane == 0) L.start[orient][ptr] = (uint8_t)pick;
            wave_fence();
            if (end >= start) lookups += (u64)(end - start + 1) * I;
        }
    }
    // seedindex: segments ordered by (sum of bucket headers, segment)  (align.cpp:476-485)
    {
        int s = 0;
        if (lane < nseg) {
            if (P.rrbs) s = (int)cnt[BL.prof[lane][0] + (orient ? (M.u->len % S) : 0) + L.start[orient][lane]];  // GenerateCSeeds adds cseed_offset
            else
                for (int ph = 0; ph < I; ph++) s += (int)cnt[BL.prof[lane][ph] + L.start[orient][lane] - ph];
        }
        int rank = 0;
        for (int j = 0; j < nseg; j++) {
            const int sj = (int)rl((uint32_t)s, j);
            rank += (sj < s) || (sj == s && j < lane);
        }
        if (lane < nseg) L.order[orient][rank] = (uint8_t)lane;
        wave_fence();
        lookups += P.rrbs ? (u64)nseg : (u64)nseg * I;
    }
    C.n_lookup += lookups;
}

// RefSeq::int2hit (dbseq.cpp:585-595) on the LDS copy of the anchors (global memory when there are too many)
__device__ __forceinline__ uint32_t chr_of(const DevParams &P, const BlockLds &BL, uint32_t p)
{
    int left = 0, right = (int)P.n_chr;
    const bool lds = P.n_chr <= BSX_LDS_CHR;
    while (left < right - 1) {
        const int mid = (left + right) >> 1;
        const uint32_t a = lds ? BL.anchor[mid] : P.anchor[mid];
        if (p >= a) left = mid; else right = mid;
    }
    return (uint32_t)left;
}

// duplicate test against every hit accepted so far for this read (hitset, align.cpp:274).  A 4096-bit filter held in
// two VGPRs answers "certainly new" without touching memory; only on a filter hit are the exact keys compared
// (first 64 in a register, the rest in the wave's HBM slab, 256 per round trip).
__device__ __forceinline__ uint32_t bloom_slot(uint32_t key) { return (key * 0x9E3779B1u) >> 20; }  // 12 bits

__device__ __forceinline__ uint32_t hset_home(uint32_t key, uint32_t hbits) { return (key * 0x85EBCA6Bu) >> (32 - hbits); }

__device__ __forceinline__ bool seen_before(const Mate &M, const Slab &SL, uint32_t key, int lane)
{
    const uint32_t slot = bloom_slot(key);
    const uint32_t word = (slot & 2048) ? rl(M.bloom1, (slot >> 5) & 63) : rl(M.bloom0, (slot >> 5) & 63);
    if (!((word >> (slot & 31)) & 1)) return false;
    if (bsx_ballot((uint32_t)lane < min(M.u->nkeys, 64u) && M.key_reg == key)) return true;
    if (M.u->nkeys <= 64) return false;
    // linear probing, 64 slots per step: found if the key shows up before the first empty slot
    const uint32_t hmask = (1u << SL.hbits) - 1;
    for (uint32_t h = hset_home(key, SL.hbits);; h = (h + 64) & hmask) {
        const uint32_t v = SL.hset[(h + lane) & hmask];
        const u64 hit = bsx_ballot(v == key + 1), empty = bsx_ballot(v == 0);
        if (hit && (!empty || __builtin_ctzll(hit) < __builtin_ctzll(empty))) return true;
        if (empty) return false;
    }
}

__device__ __forceinline__ void remember_key(Mate &M, const Slab &SL, uint32_t key, int lane)
{
    if (M.u->nkeys < 64) { if ((uint32_t)lane == M.u->nkeys) M.key_reg = key; }
    else if (M.u->nkeys >= SL.kcap) { M.u->flags |= 4u; return; }  // set full (RRBS only, see Slab): the coordinate is not remembered, the unit is flagged (bit 2 of flags)
    else {
        const uint32_t hmask = (1u << SL.hbits) - 1;
        for (uint32_t h = hset_home(key, SL.hbits);; h = (h + 64) & hmask) {
            const uint32_t sidx = (h + lane) & hmask;
            const u64 empty = bsx_ballot(SL.hset[sidx] == 0);
            if (empty) {
                if (lane == (int)__builtin_ctzll(empty)) { SL.hset[sidx] = key + 1; SL.keys[M.u->nkeys] = key; SL.kslot[M.u->nkeys] = sidx; }
                break;
            }
        }
        wave_fence();
    }
    M.u->nkeys++;
    const uint32_t slot = bloom_slot(key);
    // (both words are written unconditionally: an "if / else" over two fields becomes a store through a selected address, and
    //  one such store keeps the whole Mate in private memory — every M.field access a scratch round trip — instead of registers)
    const uint32_t bit = (uint32_t)lane == ((slot >> 5) & 63) ? 1u << (slot & 31) : 0u;
    M.bloom0 |= (slot & 2048) ? 0u : bit;
    M.bloom1 |= (slot & 2048) ? bit : 0u;
}

// leave the hash set empty for the next unit that uses this slab
__device__ __forceinline__ void forget_keys(const Mate &M, const Slab &SL, int lane)
{
    for (uint32_t i = 64 + lane; i < M.u->nkeys; i += 64) SL.hset[SL.kslot[i]] = 0;
    wave_fence();
}

// ---------------------------------------------------------------------------------------------------------------
// SnpAlign (align.cpp:168-347) — returns when the reference's SnpAlign would return
// ---------------------------------------------------------------------------------------------------------------
// RefSeq::CCGG_seglen (dbseq.cpp:541-567), fragment length only
// BINS: start the search from the 4 kb bin table.  Only the control kernel of the heavy pipeline does (where it replays hundreds of
// survivors per read); the main kernel keeps the plain search — its register allocation sits at the 128-VGPR limit and the
// extra code there costs the WGBS instantiation 1 KB of spill per lane.
template <bool BINS>
__device__ int ccgg_seglen(const DevParams &P, uint32_t chr, uint32_t pos, int readlen)
{
    const uint32_t c = chr >> 1;
    const uint32_t *sites = P.sites + P.site_off[c];
    const int size = (int)(P.site_off[c + 1] - P.site_off[c]);
    int left = 0, right = size - 1;
    if (BINS && P.site_bin && size >= 2) {
        // The reference's loop (below) ends with left = the last site <= pos, kept inside [0, size-2], and right = left + 1 (an
        // exact hit can only be a probe strictly between the ends, which gives the same).  The same pair from the 4 kb bin table:
        // a handful of sites to search instead of the whole chromosome's list (17 dependent loads on the hg38-sized genome).
        const uint32_t *tab = P.site_bin + P.site_bin_off[c];
        const uint32_t nb = P.site_bin_off[c + 1] - P.site_bin_off[c], b = min(pos >> BSX_SITE_BIN_SHIFT, nb - 2);
        int lo = (int)tab[b], hi = (pos >> BSX_SITE_BIN_SHIFT) > nb - 2 ? size : (int)tab[b + 1];  // sites before lo are < the bin, sites from hi on are beyond it
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (sites[mid] <= pos) lo = mid + 1; else hi = mid; }  // lo = number of sites <= pos
        left = max(0, min(size - 2, lo - 1)); right = left + 1;
    } else
    while (left < right - 1) {
        const int mid = (left + right) / 2;
        const uint32_t mv = sites[mid];
        if (mv == pos) { left = mid; right = mid + 1; break; }
        else if (mv < pos) left = mid;
        else right = mid;
    }
    const uint32_t seg_start = size > 0 ? sites[left] : 0;
    uint32_t seg_end;
    for (;;) {
        // the reference reads sites[right] before testing right < size; the one-past-the-end value is defined as 0 here
        const uint32_t sv = (right >= 0 && right < size) ? sites[right] : 0;
        seg_end = sv + (uint32_t)P.digest_len - (uint32_t)(P.digest_pos * 2);
        if (seg_end < pos + (uint32_t)readlen && right < size) right++;
        else break;
    }
    return (int)(seg_end - seg_start);
}

struct CandEval { uint32_t w, w0ref, p48, w01ref; };

// CountMismatch (align.h:167-200) for the candidate starting at global nt p of one strand copy.  The reference
// compares a pre-shifted copy of the read with aligned reference words; here the reference words are funnel-shifted
// into the read's frame instead.  w0ref / w01ref are the partial sums the reference's two early-outs look at.
__device__ __forceinline__ CandEval eval_loaded(const uint32_t *rp, const U4 r0, const uint32_t (&rw)[9], const uint32_t (&rm)[9], int nwords,
                                                uint32_t p, uint32_t thres0)
{
    // rp / r0 start at word (p-1)>>4: one word early when p is word-aligned, so that the funnel shift is a plain
    // v_alignbit_b32 with a 5-bit amount (k = 0 -> amount 0 picks the second word of each pair)
    CandEval r;
    const uint32_t k = p & 15, sh = (32 - 2 * k) & 31;
    const uint32_t f0 = __builtin_amdgcn_alignbit(r0.a, r0.b, sh), f1 = __builtin_amdgcn_alignbit(r0.b, r0.c, sh),
                   f2 = __builtin_amdgcn_alignbit(r0.c, r0.d, sh);
    const uint32_t m1 = bsx_mismatch_hi(rw[1], bsx_tmask(rw[1], rm[1]), f1);
    const uint32_t c0 = __popc(bsx_mismatch_hi(rw[0], bsx_tmask(rw[0], rm[0]), f0));
    const uint32_t him = k ? ~(0xFFFFFFFFu >> (2 * (16 - k))) : 0xFFFFFFFFu;  // first 16-k nt of a word
    r.w0ref = c0 + __popc(m1 & him);  // the reference's 1st 64-bit word holds read nt [0, 32-k)
    r.p48 = c0 + __popc(m1) + __popc(bsx_mismatch_hi(rw[2], bsx_tmask(rw[2], rm[2]), f2));
    r.w = r.p48;
    r.w01ref = 0;
    if (r.p48 <= thres0) {
        uint32_t wd[7];
        wd[0] = r0.d;
        if (nwords > 3) { const U4 r1 = ldm4(rp + 4); wd[1] = r1.a; wd[2] = r1.b; wd[3] = r1.c; wd[4] = r1.d; }
        else { wd[1] = wd[2] = wd[3] = wd[4] = 0; }
        if (nwords > 7) { const U2 r2 = ldm2(rp + 8); wd[5] = r2.a; wd[6] = r2.b; }
        else { wd[5] = wd[6] = 0; }
        uint32_t tot = r.p48;
        r.w01ref = r.p48;
#pragma unroll
        for (int t = 3; t < 9; t++) {
            const uint32_t f = __builtin_amdgcn_alignbit(wd[t - 3], wd[t - 2], sh);
            const uint32_t mm = bsx_mismatch_hi(rw[t], bsx_tmask(rw[t], rm[t]), f);
            tot += __popc(mm);
            if (t == 3) r.w01ref += __popc(mm & him);
        }
        r.w = tot;
    }
    return r;
}

__device__ __forceinline__ CandEval eval_candidate(const DevParams &P, const uint32_t (&rw)[9], const uint32_t (&rm)[9], int nwords,
                                                   uint32_t p, uint32_t strand, uint32_t thres0)
{
    const uint32_t *rp = (strand ? P.crefcat : P.refcat) + ((p - 1) >> 4);
    const U4 r0 = ldm4(rp);
    return eval_loaded(rp, r0, rw, rm, nwords, p, thres0);
}

// the same evaluation in two steps, for k_hscan: the first 48 nt decide for most candidates, and the words behind them
// are requested for all four chunks of a step at once instead of chunk by chunk
struct HeadEval { uint32_t p48, w0ref; };
__device__ __forceinline__ HeadEval eval_head(const U4 r0, const uint32_t (&rw)[9], const uint32_t (&rm)[9], uint32_t p)
{
    HeadEval r;
    const uint32_t k = p & 15, sh = (32 - 2 * k) & 31;
    const uint32_t f0 = __builtin_amdgcn_alignbit(r0.a, r0.b, sh), f1 = __builtin_amdgcn_alignbit(r0.b, r0.c, sh),
                   f2 = __builtin_amdgcn_alignbit(r0.c, r0.d, sh);
    const uint32_t m1 = bsx_mismatch_hi(rw[1], bsx_tmask(rw[1], rm[1]), f1);
    const uint32_t c0 = __popc(bsx_mismatch_hi(rw[0], bsx_tmask(rw[0], rm[0]), f0));
    const uint32_t him = k ? ~(0xFFFFFFFFu >> (2 * (16 - k))) : 0xFFFFFFFFu;
    r.w0ref = c0 + __popc(m1 & him);
    r.p48 = c0 + __popc(m1) + __popc(bsx_mismatch_hi(rw[2], bsx_tmask(rw[2], rm[2]), f2));
    return r;
}
// words 3..8: d = fourth word of the first load, r1 / r2 = the six words behind it (zero where the read is shorter)
__device__ __forceinline__ void eval_tail(uint32_t d, const U4 r1, const U2 r2, const uint32_t (&rw)[9], const uint32_t (&rm)[9], uint32_t p, uint32_t p48,
                                          uint32_t &w, uint32_t &w01ref)
{
    const uint32_t k = p & 15, sh = (32 - 2 * k) & 31;
    const uint32_t him = k ? ~(0xFFFFFFFFu >> (2 * (16 - k))) : 0xFFFFFFFFu;
    const uint32_t wd[7] = {d, r1.a, r1.b, r1.c, r1.d, r2.a, r2.b};
    uint32_t tot = p48;
    w01ref = p48;
#pragma unroll
    for (int t = 3; t < 9; t++) {
        const uint32_t f = __builtin_amdgcn_alignbit(wd[t - 3], wd[t - 2], sh);
        const uint32_t mm = bsx_mismatch_hi(rw[t], bsx_tmask(rw[t], rm[t]), f);
        tot += __popc(mm);
        if (t == 3) w01ref += __popc(mm & him);
    }
    w = tot;
}

// hit coordinates of a WGBS candidate at global nt p of strand copy `strand`; false if it runs off the chromosome
__device__ __forceinline__ bool hit_coords(const DevParams &P, const BlockLds &BL, uint32_t p, uint32_t strand, int len, uint32_t &hchr,
                                           uint32_t &hloc, uint32_t &hkey)
{
    const bool lds_chr = P.n_chr <= BSX_LDS_CHR;
    const uint32_t c = chr_of(P, BL, p);
    const uint32_t an = lds_chr ? BL.anchor[c] : P.anchor[c], sz = lds_chr ? BL.chr_size[c] : P.chr_size[c];
    uint32_t loc = p - an;
    if (strand) loc = (lds_chr ? BL.rc_offset[c] : P.rc_offset[c]) - (uint32_t)len - loc;  // align.cpp:289
    hchr = 2 * c + strand; hloc = loc; hkey = an + loc;
    return !((u64)loc + (u64)len > (u64)sz);  // overflow the end of refseq (align.cpp:273)
}

// One accepted-candidate step of the reference's inner loops (align.cpp:274-278 and the RRBS twin :201-212):
// hitset.insert, optional fragment filter, append, -r 0 early return, -w cap.  Returns 0 nothing happened /
// 1 threshold lowered / 2 SnpAlign returns.
template <bool BINS>
__device__ __forceinline__ int accept_survivor(const DevParams &P, Mate &M, const Slab &SL, int orient, int mode, uint32_t ws, uint32_t hchr,
                                               uint32_t hloc, uint32_t hkey, int lane)
{
    if (ws > M.u->snp_thres) return 0;
    if (seen_before(M, SL, hkey, lane)) return 0;
    remember_key(M, SL, hkey, lane);  // hitset.insert
    if (P.rrbs && !P.pairend && orient == 0) {  // fragment size filter, forward chain only (align.cpp:202-207)
        const int sl = ccgg_seglen<BINS>(P, hchr, hloc, M.u->len);
        if (sl > P.max_insert || sl < P.min_insert) return 0;
    }
    const uint32_t n = n_of(M, orient, (int)ws);
    if (lane == 0) SL.list(orient, (int)ws)[n] = ((u64)hchr << 32) | hloc;  // hits[w][n++] = hit
    if (lane == orient * 16 + (int)ws) M.cnt_reg++;
    const uint32_t both = n_of(M, 0, (int)ws) + n_of(M, 1, (int)ws);
    if ((int)ws == mode && !P.pairend && P.report_repeat_hits == 0 && both > 1) return 2;
    if (both >= (uint32_t)P.max_num_hits) {
        if (ws == 0) return 2;
        M.u->snp_thres = ws - 1;
        return 1;
    }
    return 0;
}

// accept_survivor for up to 64 survivors of one list at once — lane l holds survivor l of `act`, in list order.  Used
// where a read collects hundreds of hits (the heavy pipeline): the one-at-a-time form pays a dependent memory round
// trip or two per survivor, this one a handful per 64.  It stops behind the first survivor that causes an event, exactly
// where the sequential loop would: that survivor is committed, the ones behind it are untouched, and its lane is
// returned in ev_lane (return value as accept_survivor: 0 none / 1 threshold lowered / 2 SnpAlign returns).
#define BSX_GROUP_MIN 6  /* fewer survivors than this are cheaper one at a time */
__device__ int accept_group(const DevParams &P, Mate &M, const Slab &SL, int orient, int mode, u64 act, uint32_t ws, uint32_t hchr, uint32_t hloc,
                            uint32_t hkey, int lane, int &ev_lane)
{
    ev_lane = -1;
    bool cand = ((act >> lane) & 1) && ws <= M.u->snp_thres;
    // already in the hitset?  filter bits live in other lanes' registers: fetch the word, then the exact tests
    {
        const uint32_t slot = bloom_slot(hkey);
        const int src = (int)((slot >> 5) & 63) * 4;
        const uint32_t w0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)M.bloom0), w1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)M.bloom1);
        const bool maybe = cand && ((((slot & 2048) ? w1 : w0) >> (slot & 31)) & 1);
        if (bsx_ballot(maybe)) {
            bool found = false;
            const uint32_t nk = min(M.u->nkeys, 64u);
            for (uint32_t j = 0; j < nk; j++) found |= rl(M.key_reg, (int)j) == hkey;
            bool probing = maybe && !found && M.u->nkeys > 64;
            uint32_t h = hset_home(hkey, SL.hbits);
            while (bsx_ballot(probing)) {
                if (probing) {
                    const uint32_t v = SL.hset[h];
                    if (v == hkey + 1) { found = true; probing = false; }
                    else if (v == 0) probing = false;
                    else h = (h + 1) & ((1u << SL.hbits) - 1);
                }
            }
            if (maybe && found) cand = false;
        }
    }
    // the same coordinate twice inside the group: the first occurrence wins
    for (u64 t = bsx_ballot(cand); t; t &= t - 1) {
        const int j = (int)__builtin_ctzll(t);
        if (lane > j && rl(hkey, j) == hkey) cand = false;
    }
    if (!bsx_ballot(cand)) return 0;
    // RRBS single-end, forward chain: a new coordinate is remembered first and then dropped if its restriction fragment is
    // out of range (align.cpp:201-207) — `cand` lanes enter the hitset, only `app` lanes are appended and can cause events
    bool app = cand;
    if (P.rrbs && !P.pairend && orient == 0 && cand) {
        const int sl = ccgg_seglen<true>(P, hchr, hloc, M.u->len);
        app = !(sl > P.max_insert || sl < P.min_insert);
    }
    // position inside the class list and the first event
    uint32_t rank = 0, mine = 0, other = 0;
    const uint32_t cmax = min(M.u->snp_thres, (uint32_t)BSX_MAXSNPS);
    for (uint32_t c = 0; c <= cmax; c++) {
        const u64 mc = bsx_ballot(app && ws == c);
        if (!mc) continue;
        const uint32_t a = n_of(M, orient, (int)c), b = n_of(M, 1 - orient, (int)c);
        if (app && ws == c) { rank = (uint32_t)__builtin_popcountll(mc & (lanemask_lt(lane) | (1ull << lane))); mine = a; other = b; }
    }
    const uint32_t both = mine + other + rank;
    const bool ev2 = app && (((int)ws == mode && !P.pairend && P.report_repeat_hits == 0 && both > 1) || (both >= (uint32_t)P.max_num_hits && ws == 0));
    const bool ev1 = app && !ev2 && both >= (uint32_t)P.max_num_hits;
    const u64 em2 = bsx_ballot(ev2), em = em2 | bsx_ballot(ev1);
    const int E = em ? (int)__builtin_ctzll(em) : 64;
    const bool commit = app && lane <= E, commit_key = cand && lane <= E;
    const u64 km = bsx_ballot(commit_key);
    if (commit) SL.list(orient, (int)ws)[mine + rank - 1] = ((u64)hchr << 32) | hloc;  // hits[w][n++] = hit
    for (uint32_t c = 0; c <= cmax; c++) {
        const u64 mc = bsx_ballot(commit && ws == c);
        if (mc && lane == orient * 16 + (int)c) M.cnt_reg += (uint32_t)__builtin_popcountll(mc);
    }
    // hitset.insert: registers for the first 64 coordinates, the slab's hash set beyond, filter bits for all
    const uint32_t kidx = M.u->nkeys + (uint32_t)__builtin_popcountll(km & lanemask_lt(lane));
    {
        uint32_t k = M.u->nkeys;
        for (u64 t = km; t; t &= t - 1, k++) {
            const int j = (int)__builtin_ctzll(t);
            const uint32_t kv = rl(hkey, j), slot = bloom_slot(kv);
            if (k < 64 && (uint32_t)lane == k) M.key_reg = kv;
            const uint32_t bit = (uint32_t)lane == ((slot >> 5) & 63) ? 1u << (slot & 31) : 0u;
            M.bloom0 |= (slot & 2048) ? 0u : bit;  // (no store through a selected address: see remember_key)
            M.bloom1 |= (slot & 2048) ? bit : 0u;
        }
    }
    {
        if (bsx_ballot(commit_key && kidx >= SL.kcap)) M.u->flags |= 4u;  // set full: the unit is flagged and redone (see k_hctrl); nothing is written past the arrays
        bool pending = commit_key && kidx >= 64 && kidx < SL.kcap;
        uint32_t h = hset_home(hkey, SL.hbits);
        while (bsx_ballot(pending)) {  // claim by write-then-verify: lanes racing for one empty slot see who landed
            if (pending && SL.hset[h] == 0) SL.hset[h] = hkey + 1;
            wave_fence();
            if (pending) {
                if (SL.hset[h] == hkey + 1) { pending = false; SL.keys[kidx] = hkey; SL.kslot[kidx] = h; }
                else h = (h + 1) & ((1u << SL.hbits) - 1);
            }
            wave_fence();
        }
    }
    M.u->nkeys = min(M.u->nkeys + (uint32_t)__builtin_popcountll(km), max(SL.kcap, 64u));
    wave_fence();
    if (!em) return 0;
    ev_lane = E;
    if ((em2 >> E) & 1) return 2;
    M.u->snp_thres = rl(ws, E) - 1;
    return 1;
}

// The candidate list of one SnpAlign call for one read orientation.  WGBS: the (phase x strand) sub-ranges of the
// index, sub-range s = 2*phase+strand described by lane s (align.cpp:258-299); RRBS: one bucket of {tag,loc} pairs
// (align.cpp:175-252).  Candidates are numbered 0..total-1 in the order the reference visits them.
struct CandList {
    uint32_t sub_base, sub_n, sub_h, sub_pre;  // lane s: first entry, size, h, first ordinal of sub-range s
    uint32_t total;
    int nsub;
};

template <bool EXACT>
__device__ __forceinline__ CandList make_list(const DevParams &P, const BlockLds &BL, const MateLds &L, const Mate &M, int orient, int seg, int lane)
{
    CandList cl;
    cl.sub_base = cl.sub_n = cl.sub_h = 0;
    cl.nsub = P.rrbs ? 1 : 2 * P.index_interval;
    if (lane < cl.nsub) {
        if (P.rrbs) {
            const int a = BL.prof[seg][0];
            const int coff = orient ? (M.u->len % P.seed_size) : 0;  // cseed_offset (align.cpp:443)
            const uint32_t key = seed_key_at(P, L.w[orient], a + coff + L.start[orient][seg]);
            // the entries of this read's (segment, direction) are a contiguous part of the bucket (bsx_index_build_rrbs);
            // a group outside 0..15 has no entry (the reference's tag filter rejects the whole bucket)
            const uint32_t g_ = orient ? (uint32_t)(M.u->nfull - 1 - seg) : (uint32_t)seg;  // cmodeindex (align.cpp:221) / modeindex
            const U2 b = *reinterpret_cast<const U2 *>(P.rrbs_goff + (size_t)key * 32 + (orient << 4) + (g_ & 15));
            cl.sub_base = b.a; cl.sub_n = g_ > 15u ? 0u : b.b - b.a;
            cl.sub_h = (uint32_t)(a + coff);
        } else {
            const int ph = lane >> 1;
            const int a = BL.prof[seg][ph], st = L.start[orient][seg];
            const uint32_t key = key_at<EXACT>(P, L, M, orient, a + st - ph);
            const U2 b = ldm2(P.bucket_off + key);
            const uint32_t nf = ldm1(P.bucket_nfwd + key);
            cl.sub_base = (lane & 1) ? b.a + nf : b.a;
            cl.sub_n = (lane & 1) ? (b.b - b.a - nf) : nf;
            cl.sub_h = (uint32_t)(-a + ph - st);  // h (align.cpp:263)
        }
    }
    uint32_t pre = cl.sub_n;  // inclusive prefix over lanes
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) { uint32_t t = __shfl_up(pre, o); if (lane >= o) pre += t; }
    cl.total = rl(pre, cl.nsub - 1);
    cl.sub_pre = pre - cl.sub_n;
    return cl;
}

// Scan candidates [c_begin, c_end) of a list with one wave, 64 at a time, replaying survivors in order.
//   COUNT_ONLY: no replay; only the work counters are advanced, with the threshold frozen at thres_fixed (used to
//   re-count the part of a pre-scanned task that precedes an event).
// Returns 0 = range finished, 1 = range finished and the threshold was lowered on the way, 2 = SnpAlign returns.
// PE: instantiated for a paired batch (only picks the form of the single-end RRBS fragment filter's site search, see ccgg_seglen)
// CTX (round 5, the main kernel with the work counters off): the index keeps, beside every entry, the 32 reference nt left and right of its seed
// (DevParams::ctx, 16 bytes per entry, bsx_index.hip).  A candidate is first compared with the read's own flanks of that seed — data that comes with the
// coalesced entry load — and only candidates within the threshold THERE gather their reference words: a candidate that fails on a part of the read fails on
// the whole (the prefilter only ever removes candidates the full count would reject; the threshold can only fall while a list is walked), so every hit list is
// unchanged, and four candidates in five never touch the reference.  The work counters need every candidate's first-word count: they take the plain path.
template <bool COUNT_ONLY, int BSX_SCAN_NB, bool PE = true, bool CTX = false>
__device__ int wave_scan_range(const DevParams &P, const BlockLds &BL, const MateLds &L, Mate &M, const Slab &SL, const CandList &cl, int orient,
                               int seg, int mode, uint32_t c_begin, uint32_t c_end, uint32_t thres_fixed, int lane, Counters &C)
{
    const int nwords = (M.u->len + 15) >> 4;
    const bool lds_chr = P.n_chr <= BSX_LDS_CHR;
    uint32_t rw[9], rm[9];  // read words + masks in scalar registers
#pragma unroll
    for (int t = 0; t < 9; t++) { rw[t] = rfl(L.w[orient][t]); rm[t] = rfl(L.m[orient][t]); }
    const int cmode = M.u->nfull - 1 - seg;  // cmodeindex (align.cpp:221)
    int status = 0;
    // BSX_SCAN_NB chunks of 64 candidates per step: all their index entries are requested together, then all their
    // first reference words, and only then are the chunks evaluated one after the other in list order (two memory
    // round trips per step instead of two per chunk; the threshold each chunk sees is still the one left by its
    // predecessors, and a chunk behind an early return is simply dropped).  The main kernel uses 1 (its lists are
    // short and its register budget is tight), the control kernel of the heavy pipeline 4.
    // WGBS, one chunk per step (the main kernel): the index entry of the NEXT chunk is requested before this chunk's reference gather, so a
    // chunk is two dependent memory round trips (gather, then the words behind 48 nt where a lane needs them) instead of three — the main
    // kernel is a chain of such round trips (SQ_WAIT_ANY 0.76 of its wave cycles).  e_pf: the prefetched entry, hs_pf: its h << 1 | strand.
    constexpr bool PREFETCH = BSX_SCAN_NB == 1 && BSX_MAIN_PREFETCH;
    // (a macro, not a lambda: a closure over `cl` put the whole list descriptor into scratch — 1.1 KB per lane)
    // (hs: h << 5 | sub-range: its lowest bit is the strand copy, the others the index phase)
    // (round 5: the sub-range of a candidate = the number of later sub-ranges that begin at or before it — their prefix sums never fall, an empty one
    // shares its begin with its successor — counted against seven scalars; its entry offset and h come from the lane that holds the sub-range.
    // 18 vector instructions per chunk where a loop over the sub-ranges with its four v_readlane each took ~80.)
    uint32_t loc_pk[7];
#pragma unroll
    for (int k_ = 0; k_ < 7; k_++) loc_pk[k_] = (PREFETCH && k_ + 1 < cl.nsub) ? rl(cl.sub_pre, k_ + 1) : 0xffffffffu;
    const uint32_t loc_delta = cl.sub_base - cl.sub_pre, loc_hs = (cl.sub_h << 5) | (uint32_t)(lane & 31);
#define BSX_LOCATE(idx_, e_idx_, hs_) do { uint32_t s_ = 0; \
        _Pragma("unroll") for (int k_ = 0; k_ < 7; k_++) s_ += (idx_) >= loc_pk[k_]; \
        for (int k_ = 8; k_ < cl.nsub; k_++) s_ += (idx_) >= rl(cl.sub_pre, k_);   /* (-I above 4: up to 32 sub-ranges) */ \
        const uint32_t d_ = (uint32_t)__shfl((int)loc_delta, (int)s_);   /* (by every lane, before the select: a lane that is switched off hands out zero) */ \
        hs_ = (uint32_t)__shfl((int)loc_hs, (int)s_); e_idx_ = (idx_) < cl.total ? (idx_) + d_ : 0u; } while (0)
    constexpr bool USE_CTX = CTX && PREFETCH && !COUNT_ONLY;
    uint32_t *flt = const_cast<uint32_t *>(&L.fl[0][0]);
    if (USE_CTX) {   // the read's flanks of this list's seed, per index phase (at most four: the context is only built for -I <= 4): lane 4 ph + j -> word j of phase ph
        if (lane < 16) {
            const int ph = lane >> 2, j = lane & 3;
            const int o = -__shfl((int)cl.sub_h, min(2 * ph, cl.nsub - 1));      // the seed's offset in the read (lane s of the list descriptor holds sub-range s)
            const int x = o + (j < 2 ? 16 * j - 32 : 16 * j - 16);               // first nt of flank word j: [o - 32, o) and [o + 16, o + 48)
            const int i = x >> 4, sh = x & 15;
            const uint32_t *rw_ = L.w[orient], *rm_ = L.m[orient];
            const uint32_t w0 = (i >= 0 && i < 10) ? rw_[i] : 0u, w1 = (i + 1 >= 0 && i + 1 < 10) ? rw_[i + 1] : 0u;
            const uint32_t m0 = (i >= 0 && i < 10) ? rm_[i] : 0u, m1 = (i + 1 >= 0 && i + 1 < 10) ? rm_[i + 1] : 0u;
            const uint32_t f = sh ? __builtin_amdgcn_alignbit(w0, w1, 32u - 2u * (uint32_t)sh) : w0, mk = sh ? __builtin_amdgcn_alignbit(m0, m1, 32u - 2u * (uint32_t)sh) : m0;
            flt[ph * 8 + j] = f; flt[ph * 8 + 4 + j] = bsx_tmask(f, mk);
        }
        wave_fence();
    }
    uint32_t e_pf = 0, hs_pf = 0;
    U4 c_pf; c_pf.a = c_pf.b = c_pf.c = c_pf.d = 0;
    if (PREFETCH && !P.rrbs && c_begin < c_end) {
        uint32_t ei; const uint32_t i0_ = c_begin + (uint32_t)lane; BSX_LOCATE(i0_, ei, hs_pf); e_pf = ldm1(P.entries + ei);
        if (USE_CTX) c_pf = ldm4(P.ctx + 4 * (size_t)ei);
    }
    for (uint32_t cs = c_begin; cs < c_end; cs += 64 * BSX_SCAN_NB) {
      uint32_t p_[BSX_SCAN_NB], aux_[BSX_SCAN_NB];  // aux: WGBS strand / RRBS chromosome id
      bool valid_[BSX_SCAN_NB];
      U4 r0_[BSX_SCAN_NB];
#pragma unroll
      for (int u = 0; u < BSX_SCAN_NB; u++) {
        const uint32_t idx = cs + 64 * u + lane;
        bool valid = idx < c_end;
        uint32_t p = 16, strand = 0, rchr = 0;
        if (P.rrbs) {
            if (valid) {
                const U2 e = *reinterpret_cast<const U2 *>(P.entries + 2 * (size_t)(rl(cl.sub_base, 0) + idx));
                const uint32_t h = rl(cl.sub_h, 0);
                const bool tag_ok = orient ? (((e.a ^ 0x1000000u) >> 16) == (uint32_t)cmode) : ((e.a >> 16) == (uint32_t)seg);
                rchr = e.a & 0xffff;
                valid = tag_ok && e.b >= h;  // mode or strand not match / underflow the start of refseq
                strand = rchr & 1;
                if (valid) p = (lds_chr ? BL.anchor[rchr >> 1] : P.anchor[rchr >> 1]) + (e.b - h);
            }
        } else if (PREFETCH) {
            const uint32_t e = e_pf, hs = hs_pf;
            const U4 cx = c_pf;
            strand = hs & 1u;
            if (cs + 64 < c_end) {   // (a lane behind the list's end reads entry 0: never used)
                uint32_t ei; const uint32_t in_ = idx + 64u; BSX_LOCATE(in_, ei, hs_pf); e_pf = ldm1(P.entries + ei);
                if (USE_CTX) c_pf = ldm4(P.ctx + 4 * (size_t)ei);
            }
            if (USE_CTX && valid) {   // the read's flanks of the seed against the entry's context: within the threshold there, or no candidate at all
                const uint4 *fr = reinterpret_cast<const uint4 *>(flt + ((hs >> 1) & 3u) * 8u);
                const uint4 f = fr[0], t = fr[1];
                const uint32_t cn = __popc(bsx_mismatch_hi(f.x, t.x, cx.a)) + __popc(bsx_mismatch_hi(f.y, t.y, cx.b)) + __popc(bsx_mismatch_hi(f.z, t.z, cx.c)) + __popc(bsx_mismatch_hi(f.w, t.w, cx.d));
                valid = cn <= M.u->snp_thres;
            }
            if (valid) p = e + (uint32_t)((int32_t)hs >> 5);
        } else {
            uint32_t e_idx = 0, h = 0;
            for (int s = 0; s < cl.nsub; s++) {
                const uint32_t ps = rl(cl.sub_pre, s), ns = rl(cl.sub_n, s);
                if (idx >= ps && idx < ps + ns) { e_idx = rl(cl.sub_base, s) + (idx - ps); h = rl(cl.sub_h, s); strand = s & 1; }
            }
            const uint32_t e = ldm1(P.entries + e_idx);
            if (valid) p = e + h;
        }
        p_[u] = p; aux_[u] = P.rrbs ? rchr : strand; valid_[u] = valid;
      }
      if (USE_CTX && BSX_SCAN_NB == 1 && !bsx_ballot(valid_[0])) continue;   // no candidate of the chunk got past its context: nothing to gather, nothing to replay
#pragma unroll
      for (int u = 0; u < BSX_SCAN_NB; u++)
        r0_[u] = ldm4(((P.rrbs ? (aux_[u] & 1) : aux_[u]) ? P.crefcat : P.refcat) + ((p_[u] - 1) >> 4));
#pragma unroll
      for (int u = 0; u < BSX_SCAN_NB; u++) {
        const uint32_t c0 = cs + 64 * u;
        if (c0 >= c_end) break;
        const bool valid = valid_[u];
        const uint32_t p = p_[u], rchr = aux_[u], strand = P.rrbs ? (aux_[u] & 1) : aux_[u];
        CandEval ev = {0xffff, 0, 0, 0};
        const uint32_t thres0 = COUNT_ONLY ? thres_fixed : M.u->snp_thres;
        if (valid) ev = eval_loaded((strand ? P.crefcat : P.refcat) + ((p - 1) >> 4), r0_[u], rw, rm, nwords, p, thres0);
        uint32_t thr_eff = thres0;
        bool alive = valid, stop = false;
        if (!COUNT_ONLY) {
            const uint32_t w = ev.w;
            bool pass = valid && w <= thres0;  // per-lane hit coordinates for lanes that pass the current threshold
            uint32_t hchr = 0, hloc = 0, hkey = 0;
            if (pass) {
                if (P.rrbs) {
                    const uint32_t c = rchr >> 1;
                    const uint32_t an = lds_chr ? BL.anchor[c] : P.anchor[c], sz = lds_chr ? BL.chr_size[c] : P.chr_size[c];
                    uint32_t loc = p - an;
                    if (strand) loc = (lds_chr ? BL.rc_offset[c] : P.rc_offset[c]) - (uint32_t)M.u->len - loc;
                    hchr = 2 * c + strand; hloc = loc; hkey = an + loc;
                    if ((u64)loc + (u64)M.u->len > (u64)sz) pass = false;
                } else pass = hit_coords(P, BL, p, strand, M.u->len, hchr, hloc, hkey);
            }
            u64 surv_m = bsx_ballot(pass);  // ordered replay of the survivors
            if (BSX_SCAN_NB > 1 && __builtin_popcountll(surv_m) > BSX_GROUP_MIN) {  // heavy pipeline: 64 at a time, resuming behind every threshold change
                while (surv_m) {
                    int ls;
                    const int e = accept_group(P, M, SL, orient, mode, surv_m, w, hchr, hloc, hkey, lane, ls);
                    if (e == 0) break;
                    if (e == 1) { status = 1; if (lane > ls) thr_eff = M.u->snp_thres; surv_m &= ~(lanemask_lt(ls) | (1ull << ls)); }
                    else { if (lane > ls) alive = false; stop = true; break; }
                }
                surv_m = 0;
            }
            while (surv_m) {
                const int ls = (int)__builtin_ctzll(surv_m);
                surv_m &= surv_m - 1;
                const int e = accept_survivor<(BSX_SCAN_NB > 1) || !PE>(P, M, SL, orient, mode, rl(w, ls), rl(hchr, ls), rl(hloc, ls), rl(hkey, ls), lane);
                if (e == 1) { status = 1; if (lane > ls) thr_eff = M.u->snp_thres; }
                else if (e == 2) { if (lane > ls) alive = false; stop = true; break; }
            }
        }
        // work accounting exactly as the reference's CountMismatch early-outs (align.h:189-197)
        if (!USE_CTX)   // (the prefiltered scan runs with the work counters off: what it would add is not a count of anything)
        {
            const bool one = alive && ev.w0ref > thr_eff;
            const bool two = alive && !one && (ev.p48 > thr_eff || ev.w01ref > thr_eff);
            const bool five = alive && !one && !two;
            C.n_cand += (u64)__builtin_popcountll(bsx_ballot(alive));
            C.sum_w += (u64)__builtin_popcountll(bsx_ballot(one)) + 2ull * __builtin_popcountll(bsx_ballot(two)) + 5ull * __builtin_popcountll(bsx_ballot(five));
        }
        if (stop) return 2;
      }
    }
    return status;
}

#ifndef BSX_MAIN_NB
#define BSX_MAIN_NB 1  /* chunks of 64 candidates per step of the main kernel's scan */
#endif
// SnpAlign (align.cpp:168-347) in the main kernel.  A WGBS list of heavy_threshold candidates or more sets M.u->defer and
// returns: the unit is redone from scratch by the heavy pipeline, which scans such lists with the whole chip.
template <bool EXACT, bool PE, bool CTX = false>
__device__ __forceinline__ void snp_align(const DevParams &P, const BlockLds &BL, const MateLds &L, Mate &M, const Slab &SL, int mode, int lane,
                          Counters &C, uint32_t heavy_threshold)
{
    for (int orient = 0; orient < 2; orient++) {
        if (!((M.u->flags >> orient) & 1)) continue;
        const int seg = L.order[orient][mode];  // modeindex
        const CandList cl = make_list<EXACT>(P, BL, L, M, orient, seg, lane);
        if (heavy_threshold && cl.total >= heavy_threshold) { M.u->defer = 1; return; }
        int r_;   // (inlined here whatever the inliner thinks of its size: as a call it costs the main kernel 1.1 KB of stack per lane)
        [[clang::always_inline]] r_ = wave_scan_range<false, BSX_MAIN_NB, PE, CTX>(P, BL, L, M, SL, cl, orient, seg, mode, 0, cl.total, 0, lane, C);
        if (r_ == 2) { wave_fence(); return; }
    }
    wave_fence();
}

// SingleAlign::RunAlign (align.cpp:435-452) after packing/planning
template <bool EXACT, bool PE, bool CTX = false>
__device__ __forceinline__ void run_align_single(const DevParams &P, const BlockLds &BL, const MateLds &L, Mate &M, const Slab &SL, int lane, Counters &C,
                                 uint32_t heavy_threshold)
{
    for (int i = 0; i < M.u->seedseg; i++) {
        snp_align<EXACT, PE, CTX>(P, BL, L, M, SL, i, lane, C, heavy_threshold);
        if (M.u->defer) return;
        if (!P.rrbs) {
            const u64 nz = bsx_ballot(M.cnt_reg != 0 && (lane & 15) <= i && lane < 32);
            if (nz) return;
        }
    }
}

// SingleAlign::Fix_Unpaired_Short_Fragment (align.cpp:768-791): RRBS mates shorter than -m that ended up unpaired lose
// the hits whose restriction fragment is out of range; stops at the first class that still has a hit
__device__ void fix_unpaired_short_fragment(const DevParams &P, Mate &M, const Slab &SL, int lane)
{
    if (M.u->filtered || M.u->len >= P.min_insert) return;
    for (int ii = 0; ii <= M.u->max_snp; ii++) {
        for (int orient = 0; orient < 2; orient++) {
            const uint32_t n = n_of(M, orient, ii);
            u64 *lst = SL.list(orient, ii);
            uint32_t kept = 0;
            for (uint32_t base = 0; base < n; base += 64) {
                const uint32_t i = base + lane;
                const u64 h = i < n ? lst[i] : 0;
                bool keep = false;
                if (i < n) { const int sl = ccgg_seglen<false>(P, (uint32_t)(h >> 32), (uint32_t)h, M.u->len); keep = !(sl < P.min_insert || sl > P.max_insert); }
                const u64 m = bsx_ballot(keep);
                if (keep) lst[kept + (uint32_t)__builtin_popcountll(m & lanemask_lt(lane))] = h;
                kept += (uint32_t)__builtin_popcountll(m);
            }
            if (lane == orient * 16 + ii) M.cnt_reg = kept;
            wave_fence();
        }
        if (n_of(M, 0, ii) + n_of(M, 1, ii) > 0) break;
    }
}

// first non-empty class and a deterministic pick inside it (StringAlign align.cpp:610-627 / StringAlignUnpair pairs.cpp:255-275)
__device__ void select_hit(const DevParams &P, const Mate &M, const Slab &SL, bsx_hit &out, bool unpair_semantics)
{
    out.chr = 0; out.loc = 0; out.n_best = 0; out.best_class = -1;
    out.flags = (M.u->filtered ? BSX_F_FILTERED : 0) | ((M.u->flags & 4u) ? BSX_F_LIMIT : 0);
    out.len = (uint8_t)M.u->len; out.raw_len = (uint8_t)M.u->raw_len; out.max_snp = (uint8_t)M.u->max_snp; out.seedseg = (uint8_t)M.u->seedseg;
    if (M.u->filtered) return;
    int ii; uint32_t sum = 0, nf = 0;
    for (ii = 0; ii <= M.u->max_snp; ii++) {
        nf = n_of(M, 0, ii);
        if ((sum = nf + n_of(M, 1, ii)) > 0) break;
    }
    if (sum == 0) return;
    out.n_best = (uint16_t)sum; out.best_class = (int8_t)ii;
    uint32_t j = 0;
    if (!unpair_semantics || sum > 1) j = bsx_myrand(M.u->index, P.randseed) % sum;
    u64 h;
    if (j < nf) h = SL.list(0, ii)[j];
    else { h = SL.list(1, ii)[j - nf]; out.flags |= BSX_F_CHAIN; }
    out.chr = (uint32_t)(h >> 32); out.loc = (uint32_t)h;
}

// ---------------------------------------------------------------------------------------------------------------
// paired-end pieces
// ---------------------------------------------------------------------------------------------------------------
// sort(hits, hits+n, HitComp) (align.cpp:363-368): keys (chr<<32|loc) are unique inside a list, so ranks are positions
#define BSX_LDS_SORT 1024  /* elements of the per-wave LDS sort buffer of the heavy control kernel */
__device__ void sort_list(u64 *list, uint32_t n, u64 *tmp, int lane, u64 *lds = nullptr)
{
    if (n <= 1) return;
    if (n <= 64) {
        const u64 v = (uint32_t)lane < n ? list[lane] : ~0ull;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; j++) rank += rl64(v, (int)j) < v;
        if ((uint32_t)lane < n) list[rank] = v;
    } else if (lds && n <= BSX_LDS_SORT) {
        // heavy units sort lists of up to -w hits at every level: bitonic network in LDS, one wave, keys are unique
        uint32_t N = 128;
        while (N < n) N <<= 1;
        for (uint32_t i = lane; i < N; i += 64) lds[i] = i < n ? list[i] : ~0ull;
        wave_fence();
        for (uint32_t k = 2; k <= N; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t t = lane; t < N / 2; t += 64) {
                    const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                    const u64 a = lds[i], b = lds[l];
                    if ((a > b) == ((i & k) == 0)) { lds[i] = b; lds[l] = a; }
                }
                wave_fence();
            }
        for (uint32_t i = lane; i < n; i += 64) list[i] = lds[i];
    } else {
        for (uint32_t base = 0; base < n; base += 64) {
            const uint32_t i = base + lane;
            const u64 v = i < n ? list[i] : ~0ull;
            uint32_t rank = 0;
            for (uint32_t cb = 0; cb < n; cb += 64) {
                const u64 u = cb + lane < n ? list[cb + lane] : ~0ull;
                const uint32_t lim = min(64u, n - cb);
                for (uint32_t j = 0; j < lim; j++) rank += rl64(u, (int)j) < v;
            }
            if (i < n) tmp[rank] = v;
        }
        wave_fence();
        for (uint32_t i = lane; i < n; i += 64) list[i] = tmp[i];
    }
    wave_fence();
}

struct PairSlab { uint32_t *rows; uint32_t rowcap; __device__ __forceinline__ uint32_t *row(int cls) const { return rows + (size_t)cls * rowcap * 6; } };

__device__ __forceinline__ uint32_t first_chr_ge(const u64 *lst, uint32_t from, uint32_t n, uint32_t chr, bool strict, int lane)
{
    for (uint32_t base = from; base < n; base += 64) {
        const uint32_t i = base + lane;
        const uint32_t c = i < n ? (uint32_t)(lst[i] >> 32) : 0xffffffffu;
        const u64 m = bsx_ballot(i < n && (strict ? c > chr : c >= chr));
        if (m) return base + (uint32_t)__builtin_ctzll(m);
    }
    return n;
}

// PairAlign::GetPairs (pairs.cpp:34-135)
// lds (the heavy control kernel's per-wave sort buffer, or null): the b hits of the current chromosome are staged there when more than 64 and at most BSX_LDS_SORT —
// a read pair in a repeat family brings up to -w hits per class and side on ONE chromosome, and the literal loop (every a hit against every 64 b hits, from HBM) was the
// longest single span of a C5 control pass: 31 M cycles, 13 ms, for one unit (profiles/r06d_ctrl_clocks_trim.json).  Where the staged hits are sorted by position (they are,
// bar the -w overshoot entries) the b hits within the insert range of an a hit are a contiguous run: one 64-lane probe bounds it, and only the chunks that overlap it are
// evaluated — by the same per-hit test, in the same order, so the pair rows are the reference's entry for entry.
__device__ int get_pairs(const DevParams &P, const Mate &MA, const Mate &MB, const Slab &SA, const Slab &SB, const PairSlab &PS,
                         uint32_t &pcnt_reg, int na, int nb, int lane, u64 *lds = nullptr)
{
    if (na > MA.u->max_snp || nb > MB.u->max_snp) return 0;
    const int cls = na + nb;
    uint32_t cnt = rl(pcnt_reg, cls);
    uint32_t *row = PS.row(cls);
    int result = -1;
    for (int pass = 0; pass < 2 && result < 0; pass++) {
        const u64 *al = SA.list(pass, na), *bl = SB.list(1 - pass, nb);
        const uint32_t n_a = n_of(MA, pass, na), n_b = n_of(MB, 1 - pass, nb);
        uint32_t chra = 0xffffffffu, bstart = 0, bend = 0;
        u64 hb_cache = 0;     // the b hits of the current chromosome, when there are at most 64 of them
        bool cached = false, staged = false, narrow = false;
        uint32_t stride = 1;
        for (uint32_t a_base = 0; a_base < n_a && result < 0; a_base += 64) {
          // 64 a hits per load; the loops below then run out of registers (the lists live in HBM: a load per a hit and
          // per 64 b hits made this join the longest span of the heavy control passes)
          const u64 hav = a_base + lane < n_a ? al[a_base + lane] : 0;
          const uint32_t na_chunk = min(64u, n_a - a_base);
          for (uint32_t ii = 0; ii < na_chunk && result < 0; ii++) {
            const u64 ha = rl64(hav, (int)ii);
            const uint32_t achr = (uint32_t)(ha >> 32), aloc = (uint32_t)ha;
            if (chra != achr) {
                chra = achr;
                bstart = first_chr_ge(bl, bend, n_b, chra, false, lane);
                bend = first_chr_ge(bl, bstart, n_b, chra, true, lane);
                cached = bend - bstart <= 64;
                if (cached) hb_cache = bstart + lane < bend ? bl[bstart + lane] : 0;
                const uint32_t nr = bend - bstart;
                staged = lds != nullptr && !cached && nr <= (uint32_t)BSX_LDS_SORT;
                narrow = false;
                if (staged) {
                    bool okl = true;   // positions ascending and small enough for the interval arithmetic below to be exact
                    for (uint32_t i = (uint32_t)lane; i < nr; i += 64) {
                        const u64 v = bl[bstart + i];
                        lds[i] = v;
                        okl = okl && (uint32_t)v < 0x40000000u && (i + 1 >= nr || (uint32_t)v <= (uint32_t)bl[bstart + i + 1]);
                    }
                    wave_fence();
                    narrow = !bsx_ballot(!okl);
                    stride = (nr + 63u) / 64u;
                }
            }
            uint32_t j_lo = bstart, j_hi = bend;
            if (narrow && aloc < 0x40000000u) {
                // the b positions whose insert size can lie in [min_insert, max_insert] (pairs.cpp:72-75,99-102), and a run of staged hits that holds them all
                const bool odd_ = (chra & 1) != 0, b_first = pass == 0 ? odd_ : !odd_;
                const long long lo_ = b_first ? (long long)aloc + MA.u->len - P.max_insert : (long long)aloc + P.min_insert - MB.u->len;
                const long long hi_ = b_first ? (long long)aloc + MA.u->len - P.min_insert : (long long)aloc + P.max_insert - MB.u->len;
                const uint32_t nr = bend - bstart, pi = (uint32_t)lane * stride;
                const long long pv = pi < nr ? (long long)(uint32_t)lds[pi] : (1ll << 40);
                const u64 ge_lo = bsx_ballot(pv >= lo_), gt_hi = bsx_ballot(pv > hi_);
                const uint32_t f_lo = ge_lo ? (uint32_t)__builtin_ctzll(ge_lo) : 64u, f_hi = gt_hi ? (uint32_t)__builtin_ctzll(gt_hi) : 64u;
                j_lo = bstart + (f_lo ? (f_lo - 1u) * stride : 0u);
                j_hi = min(bend, bstart + f_hi * stride);
                if (j_hi < j_lo) j_hi = j_lo;
            }
            for (uint32_t jb = j_lo; jb < j_hi && result < 0; jb += 64) {
                const uint32_t j = jb + lane;
                const bool valid = j < j_hi;
                const u64 hb = cached ? hb_cache : (valid ? (staged ? lds[j - bstart] : bl[j]) : 0);
                const uint32_t bloc = (uint32_t)hb;
                uint32_t seg_start, seg_end;
                const bool odd = (chra & 1) != 0;
                if (pass == 0 ? odd : !odd) { seg_start = bloc; seg_end = aloc + (uint32_t)MA.u->len; }   // pairs.cpp:72,99
                else { seg_start = aloc; seg_end = bloc + (uint32_t)MB.u->len; }                          // pairs.cpp:73,100
                const int insert = (int)(seg_end - seg_start);
                const bool ok = valid && insert >= P.min_insert && insert <= P.max_insert;
                u64 m = bsx_ballot(ok);
                while (m) {  // appends happen one at a time in the reference, each followed by the cap test
                    const int ls = (int)__builtin_ctzll(m);
                    m &= m - 1;
                    if (lane == ls) {
                        uint32_t *o = row + (size_t)cnt * 6;
                        o[0] = (uint32_t)pass | ((uint32_t)na << 16) | ((uint32_t)nb << 24);
                        o[1] = (uint32_t)insert; o[2] = achr; o[3] = aloc; o[4] = (uint32_t)(hb >> 32); o[5] = bloc;
                    }
                    cnt++;
                    if (cnt >= (uint32_t)P.max_num_hits) { result = 1; break; }
                }
            }
          }
        }
    }
    if (lane == cls) pcnt_reg = cnt;
    wave_fence();
    if (result < 0) result = cnt > 0 ? 1 : 0;
    return result;
}

// ---------------------------------------------------------------------------------------------------------------
// per-unit set-up and result record (shared by the main kernel and the heavy pipeline)
// ---------------------------------------------------------------------------------------------------------------
struct UnitSlabs { Slab SA, SB; PairSlab PS; };

__device__ __forceinline__ UnitSlabs carve_slab(uint8_t *slab, uint32_t nclass, uint32_t rowcap, bool pe, uint32_t kcap, uint32_t hbits)
{
    UnitSlabs U;
    U.SA.rowcap = U.SB.rowcap = rowcap; U.SA.nclass = U.SB.nclass = nclass;
    U.SA.kcap = kcap; U.SA.hbits = hbits;
    U.SA.hits = (u64 *)slab;
    U.SA.keys = (uint32_t *)(U.SA.hits + (size_t)2 * (nclass + 1) * rowcap);
    U.SA.kslot = U.SA.keys + (size_t)kcap;
    U.SA.hset = U.SA.kslot + (size_t)kcap;
    U.SA.tmp = (u64 *)(U.SA.hset + ((size_t)1 << hbits));
    uint8_t *after_a = (uint8_t *)(U.SA.tmp + BSX_SORT_TMP);
    U.SB = U.SA;
    U.PS.rows = nullptr; U.PS.rowcap = rowcap;
    if (pe) {
        U.SB.hits = (u64 *)after_a;
        U.SB.keys = (uint32_t *)(U.SB.hits + (size_t)2 * (nclass + 1) * rowcap);
        U.SB.kslot = U.SB.keys + (size_t)kcap;
        U.SB.hset = U.SB.kslot + (size_t)kcap;
        U.SB.tmp = (u64 *)(U.SB.hset + ((size_t)1 << hbits));
        U.PS.rows = (uint32_t *)(U.SB.tmp + BSX_SORT_TMP);
    }
    return U;
}

// "-p 1 exact" mode.  The reference never resets seed_start_offset / seed_array (align.h:82-91): a read with
// (len - I + 1) % S == 0 skips the loop that sets the offset (align.cpp:458-468) and plans with the value the last read of
// its stream left behind, and its start offsets then reach seed_array entries behind its own last hash — values written by
// earlier, longer reads.  Both are pure functions of the earlier reads of the stream (mate 1 and mate 2 are separate
// streams: PairAlign owns two SingleAlign objects).  A stream here = the history the caller attached, then the units of the
// batch; before its first read the state is the one the caller handed over (bsx_batch_set_leak_state) or zero (a fresh object —
// the oracle's and the bridge's zero-initialised state).
//   k_leak_meta    : FilterReads' verdict and trimmed length of every read of the stream (2 bytes each), and per block of
//                    LEAK_BLK reads whether one of them sets the offset and the longest unfiltered one
//   leak_find      : the last read before a position that sets the offset / that wrote seed_array entry e — a backward search over
//                    those 2-byte records, 64 per step, skipping whole blocks by their summaries: a few steps whatever the input
//                    (a run of a million reads that never set the offset costs each of them ~20 steps, not a walk over all of them)
//   k_leak_resolve : per leaky read, the offset of the read leak_find names (planned once more: bucket sizes, first minimum) and
//                    the tail entries from the reads that wrote them -> LeakRec for the align kernels
//   k_leak_final   : the state behind the stream's last read (bsx_batch_get_leak_state: the next batch starts from it)
#define LEAK_BLK 4096
__device__ __forceinline__ void init_block_lds(const DevParams &P, BlockLds &BL, int tid, int nthreads);   // (inlined by force: the main kernel has to stay a leaf)
#define LEAK_KEYS 160
struct LeakState { uint32_t key[2][2][LEAK_KEYS]; uint32_t so[2][2]; };  // [mate][orientation]

template <bool PE>
__global__ __launch_bounds__(256) void k_leak_meta(AlignArgs A)
{
    __shared__ MateLds LM[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const DevParams &P = A.P;
    const uint32_t n_stream = A.n_hist + A.n_units_all;
    for (uint32_t pos = blockIdx.x * 4 + wv; pos < n_stream; pos += gridDim.x * 4) {
        for (int mate = 0; mate < (PE ? 2 : 1); mate++) {
            Mate M;
            M.u = lds_mate(&LM[wv].u);
            M.u->index = 0;
            load_and_filter(A, LM[wv], M, mate, (long)pos - (long)A.n_hist, lane);
            const int len = M.u->len;
            const bool ok = !M.u->filtered;
            if (lane == 0) {
                A.leak_meta[mate][pos] = ok ? (uint16_t)len : (uint16_t)0xffff;
                if (ok) {
                    atomicMax(&A.leak_blkmax[mate][pos / LEAK_BLK], (uint32_t)(len - P.seed_size + 1));
                    if ((len - P.index_interval + 1) % P.seed_size != 0) atomicOr(&A.leak_blkset[mate][pos / LEAK_BLK], 1u);
                }
            }
            wave_fence();
        }
    }
}

// the last position j < p of mate stream `mate` whose read sets the offset (kind 0) or has more than e seed offsets, i.e. wrote
// seed_array entry e (kind 1); -1 if there is none
__device__ long leak_find(const AlignArgs &A, int mate, long p, int kind, int e, int lane)
{
    const uint16_t *meta = A.leak_meta[mate];
    const int S = A.P.seed_size, I = A.P.index_interval;
    auto pred = [&](uint32_t v) { return v != 0xffffu && (kind == 0 ? ((int)v - I + 1) % S != 0 : (int)v - S + 1 > e); };
    auto scan_back = [&](long lo, long hi) -> long {   // the last position in [lo, hi) that satisfies pred
        for (long base = hi; base > lo; base -= 64) {
            const long idx = base - 1 - lane;
            const u64 m = bsx_ballot(idx >= lo && pred(meta[idx]));
            if (m) return base - 1 - (long)__builtin_ctzll(m);
        }
        return -1;
    };
    if (p <= 0) return -1;
    const long blk_lo = (p - 1) / LEAK_BLK * LEAK_BLK;
    long j = scan_back(blk_lo, p);   // the rest of p's own block
    if (j >= 0) return j;
    for (long bb = blk_lo / LEAK_BLK; bb > 0; bb -= 64) {   // earlier blocks by their summaries, 64 per step
        const long bi = bb - 1 - lane;
        const u64 m = bsx_ballot(bi >= 0 && (kind == 0 ? A.leak_blkset[mate][bi] != 0 : (int)A.leak_blkmax[mate][bi] > e));
        if (m) { const long bsel = bb - 1 - (long)__builtin_ctzll(m); return scan_back(bsel * LEAK_BLK, (bsel + 1) * LEAK_BLK); }
    }
    return -1;
}

// what k_leak_resolve leaves for a unit's mate (exact mode): the stale tail entries and start offsets
struct LeakRec { uint32_t key[2][16]; uint8_t so[2]; uint8_t pad[6]; };

// offset of the stream's state before position p (per orientation the stream builds) -> so[]; the initial state where no read set one
template <bool PE>
__device__ void leak_offsets(const AlignArgs &A, const BlockLds &BL, MateLds &LS, int mate, long p, uint32_t flags, int lane, uint32_t (&so)[2])
{
    const DevParams &P = A.P;
    const LeakState *init = (const LeakState *)A.leak_init;
    so[0] = init ? init->so[mate][0] : 0u; so[1] = init ? init->so[mate][1] : 0u;
    Counters dummy = {0, 0, 0, 0};
    uint32_t need = flags & 3u;
    for (long j = leak_find(A, mate, p, 0, 0, lane); j >= 0 && need; j = leak_find(A, mate, j, 0, 0, lane)) {
        Mate MJ;
        MJ.u = lds_mate(&LS.u);
        MJ.u->index = 0;
        load_and_filter(A, LS, MJ, mate, j - (long)A.n_hist, lane);
        pack_read(P, BL, LS, MJ, PE ? mate + 1 : 0, lane, dummy);
        for (int orient = 0; orient < 2; orient++) {
            if (!((need >> orient) & 1)) continue;
            plan_counts<false>(P, LS, MJ, orient, lane, false);
            const int v = plan_best_offset(P, BL, LS, MJ, orient, lane);
            if (v >= 0) { so[orient] = (uint32_t)v; need &= ~(1u << orient); }   // (a read whose totals never beat the initial minimum leaves the offset alone)
        }
        wave_fence();
    }
}

// seed_array / cseed_array entries [e0, e0 + n) (n <= 64 per call, lane i = entry e0 + i) of the stream's state before position p
template <bool PE>
__device__ void leak_entries(const AlignArgs &A, const BlockLds &BL, MateLds &LS, int mate, long p, uint32_t flags, int e0, int n, int lane, uint32_t (&key)[2])
{
    const DevParams &P = A.P;
    const LeakState *init = (const LeakState *)A.leak_init;
    const int mye = e0 + lane;
    key[0] = (init && lane < n && mye < LEAK_KEYS) ? init->key[mate][0][mye] : 0u;
    key[1] = (init && lane < n && mye < LEAK_KEYS) ? init->key[mate][1][mye] : 0u;
    Counters dummy = {0, 0, 0, 0};
    int e = e0;
    long cur = p;
    while (e < e0 + n) {
        const long j = leak_find(A, mate, cur, 1, e, lane);   // the most recent read that wrote entry e wrote everything below its own end, too
        if (j < 0) break;
        Mate MJ;
        MJ.u = lds_mate(&LS.u);
        MJ.u->index = 0;
        load_and_filter(A, LS, MJ, mate, j - (long)A.n_hist, lane);
        pack_read(P, BL, LS, MJ, PE ? mate + 1 : 0, lane, dummy);
        const int noff_j = MJ.u->len - P.seed_size + 1;
        for (int orient = 0; orient < 2; orient++) {
            if (!((flags >> orient) & 1)) continue;
            if (lane < n && mye >= e && mye < noff_j) key[orient] = seed_key_at(P, LS.w[orient], mye);   // ConvertBinaySeq wrote entries [0, noff_j) (align.cpp:101-105)
        }
        wave_fence();
        e = noff_j; cur = j;
    }
}

template <bool PE>
__global__ __launch_bounds__(256) void k_leak_resolve(AlignArgs A)
{
    __shared__ BlockLds BL;
    __shared__ MateLds LSC[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    init_block_lds(A.P, BL, threadIdx.x, 256);
    __syncthreads();
    const DevParams &P = A.P;
    for (uint32_t unit = A.first_unit + blockIdx.x * 4 + wv; unit < A.n_units; unit += gridDim.x * 4) {
        for (int mate = 0; mate < (PE ? 2 : 1); mate++) {
            const long pos = (long)A.n_hist + unit;
            const uint32_t v = rfl(A.leak_meta[mate][pos]);
            if (v == 0xffffu || ((int)v - P.index_interval + 1) % P.seed_size != 0) continue;   // filtered, or a read that sets its own offset
            const int readset = PE ? mate + 1 : 0;
            const uint32_t flags = ((P.chains || readset < 2) ? 1u : 0u) | ((P.chains || readset == 2) ? 2u : 0u);  // align.cpp:93-94
            uint32_t so[2], key[2];
            leak_offsets<PE>(A, BL, LSC[wv], mate, pos, flags, lane, so);
            leak_entries<PE>(A, BL, LSC[wv], mate, pos, flags, (int)v - P.seed_size + 1, 16, lane, key);
            LeakRec *r = (LeakRec *)A.leak_rec + ((size_t)unit * 2 + mate);
            if (lane < 16) { r->key[0][lane] = key[0]; r->key[1][lane] = key[1]; }
            if (lane < 2) r->so[lane] = (uint8_t)so[lane];
        }
        wave_fence();
    }
}

// the state behind the last read of the stream: block = mate
template <bool PE>
__global__ __launch_bounds__(64) void k_leak_final(AlignArgs A, LeakState *out)
{
    __shared__ BlockLds BL;
    __shared__ MateLds LSC;
    const int lane = threadIdx.x & 63, mate = blockIdx.x;
    init_block_lds(A.P, BL, threadIdx.x, 64);
    __syncthreads();
    const DevParams &P = A.P;
    const LeakState *init = (const LeakState *)A.leak_init;
    if (!PE && mate == 1) {   // no second stream: its state passes through
        for (int i = lane; i < (int)(sizeof(out->key[1]) / 4); i += 64) (&out->key[1][0][0])[i] = init ? (&init->key[1][0][0])[i] : 0u;
        if (lane < 2) out->so[1][lane] = init ? init->so[1][lane] : 0u;
        return;
    }
    const long pos = (long)A.n_hist + A.n_units_all;
    const int readset = PE ? mate + 1 : 0;
    const uint32_t flags = ((P.chains || readset < 2) ? 1u : 0u) | ((P.chains || readset == 2) ? 2u : 0u);
    uint32_t so[2];
    leak_offsets<PE>(A, BL, LSC, mate, pos, flags, lane, so);
    if (lane < 2) out->so[mate][lane] = so[lane];
    for (int e0 = 0; e0 < LEAK_KEYS; e0 += 64) {
        uint32_t key[2];
        const int n = min(64, LEAK_KEYS - e0);
        leak_entries<PE>(A, BL, LSC, mate, pos, flags, e0, n, lane, key);
        if (lane < n) { out->key[mate][0][e0 + lane] = key[0]; out->key[mate][1][e0 + lane] = key[1]; }
    }
}

// FilterReads + ConvertBinaySeq + ReorderSeed for the mate(s) of a unit
__device__ __forceinline__ void load_leak_rec(const AlignArgs &A, MateLds &L, uint32_t unit, int mate, int lane)
{
    const LeakRec *r = (const LeakRec *)A.leak_rec + ((size_t)unit * 2 + mate);
    if (lane < 32) (&L.stale_key[0][0])[lane] = (&r->key[0][0])[lane];
    if (lane < 2) L.stale_so[lane] = r->so[lane];
    wave_fence();
}

template <bool PE, bool EXACT>
__device__ void unit_prepare(const AlignArgs &A, const BlockLds &BL, MateLds &LA, MateLds &LB, Mate &MA, Mate &MB, uint32_t unit, int lane, Counters &C)
{
    const DevParams &P = A.P;
    MA.u->index = MB.u->index = A.first_index + unit;
    load_and_filter(A, LA, MA, 0, (long)unit, lane);
    if (PE) load_and_filter(A, LB, MB, 1, (long)unit, lane);
    else MB = MA;
    if (!MA.u->filtered) {
        pack_read(P, BL, LA, MA, PE ? 1 : 0, lane, C);
        const bool lk = EXACT && A.leak_exact && !P.rrbs && (MA.u->len - P.index_interval + 1) % P.seed_size == 0;
        if (lk) load_leak_rec(A, LA, unit, 0, lane);
        for (int o = 0; o < 2; o++) if ((MA.u->flags >> o) & 1) plan_orient<EXACT>(P, BL, LA, MA, o, lane, C, lk);  // pairs.cpp:160 / align.cpp:444
    }
    if (PE && !MB.u->filtered) {
        pack_read(P, BL, LB, MB, 2, lane, C);
        const bool lk = EXACT && A.leak_exact && !P.rrbs && (MB.u->len - P.index_interval + 1) % P.seed_size == 0;
        if (lk) load_leak_rec(A, LB, unit, 1, lane);
        for (int o = 0; o < 2; o++) if ((MB.u->flags >> o) & 1) plan_orient<EXACT>(P, BL, LB, MB, o, lane, C, lk);
    }
}

// one level of PairAlign::RunAlign after both SnpAlign calls (pairs.cpp:167-171): sort class `level`, join
__device__ int pair_level_post(const DevParams &P, const Mate &MA, const Mate &MB, const UnitSlabs &U, uint32_t &pcnt_reg, int i, int lane, u64 *lds_sort = nullptr)
{
    if (i <= MA.u->max_snp) { sort_list(U.SA.list(0, i), n_of(MA, 0, i), U.SA.tmp, lane, lds_sort); sort_list(U.SA.list(1, i), n_of(MA, 1, i), U.SA.tmp, lane, lds_sort); }
    if (i <= MB.u->max_snp) { sort_list(U.SB.list(0, i), n_of(MB, 0, i), U.SB.tmp, lane, lds_sort); sort_list(U.SB.list(1, i), n_of(MB, 1, i), U.SB.tmp, lane, lds_sort); }
    int n = get_pairs(P, MA, MB, U.SA, U.SB, U.PS, pcnt_reg, i, i, lane, lds_sort);
    for (int j = 0; j < i; j++) n += get_pairs(P, MA, MB, U.SA, U.SB, U.PS, pcnt_reg, i, j, lane, lds_sort) + get_pairs(P, MA, MB, U.SA, U.SB, U.PS, pcnt_reg, j, i, lane, lds_sort);
    return n;
}

// StringAlign / StringAlignPair / StringAlignUnpair selection, result records, clean-up
template <bool PE>
__device__ void unit_finish(const AlignArgs &A, const MateLds &LA, const MateLds &LB, Mate &MA, Mate &MB, const UnitSlabs &U, uint32_t pcnt_reg, int paired,
                            uint32_t unit, int lane, u64 &n_aligned, u64 &n_aligned_pairs)
{
    const DevParams &P = A.P;
    if (!PE) {
        bsx_hit out;
        select_hit(P, MA, U.SA, out, false);
        if (lane == 0) A.hits_out[unit] = out;
        if (lane == 0 && (out.flags & BSX_F_LIMIT)) atomicAdd((u64 *)&A.counters[16], 1ull);   // the one capacity deviation from the reference (include/bsx.h): counted, so that "never seen" is a number
        if (A.cc[0] && lane < 32) { uint16_t *cc = (uint16_t *)&A.cc[0][unit]; cc[lane] = (uint16_t)MA.cnt_reg; }
        if (A.debug && lane < 32) { A.dbg_plan[(size_t)unit * 128 + lane] = LA.start[lane >> 4][lane & 15]; A.dbg_plan[(size_t)unit * 128 + 32 + lane] = LA.order[lane >> 4][lane & 15]; }
        if (out.n_best == 1 || (out.n_best > 1 && P.report_repeat_hits == 1)) n_aligned++;
        forget_keys(MA, U.SA, lane);
        return;
    }
    bsx_pair out;
    out.a_chr = out.a_loc = out.b_chr = out.b_loc = 0; out.insert = 0; out.n_pairs = 0; out.pair_class = -1; out.chain = 0;
    out.na = out.nb = 0; out.paired = (uint8_t)paired; out.unpaired_out = 1; out.pad_ = 0;
    if (paired) {  // StringAlignPair (pairs.cpp:222-242)
        for (int c = 0; c <= 2 * P.max_snp_num; c++) {
            const uint32_t n = rl(pcnt_reg, c);
            if (!n) continue;
            out.pair_class = (int8_t)c; out.n_pairs = (uint16_t)n;
            int j = -1;
            if (n == 1) j = 0;
            else if (P.report_repeat_hits == 1) j = (int)(bsx_myrand(MA.u->index, P.randseed) % n);
            if (j >= 0) {
                const uint32_t *o = U.PS.row(c) + (size_t)j * 6;
                out.chain = (uint8_t)(o[0] & 0xffff); out.na = (uint8_t)((o[0] >> 16) & 0xff); out.nb = (uint8_t)(o[0] >> 24);
                out.insert = (int32_t)o[1]; out.a_chr = o[2]; out.a_loc = o[3]; out.b_chr = o[4]; out.b_loc = o[5];
                out.unpaired_out = 0;
            }
            break;
        }
    }
    if (P.rrbs && out.unpaired_out) { fix_unpaired_short_fragment(P, MA, U.SA, lane); fix_unpaired_short_fragment(P, MB, U.SB, lane); }  // pairs.cpp:250-253
    select_hit(P, MA, U.SA, out.a, true);
    select_hit(P, MB, U.SB, out.b, true);
    if (lane == 0) A.pairs_out[unit] = out;
    if (A.cc[0] && lane < 32) { ((uint16_t *)&A.cc[0][unit])[lane] = (uint16_t)MA.cnt_reg; ((uint16_t *)&A.cc[1][unit])[lane] = (uint16_t)MB.cnt_reg; }
    if (A.npairs_out && lane < 32) A.npairs_out[(size_t)unit * 32 + lane] = (uint16_t)pcnt_reg;
    if (A.debug && lane < 32) {
        A.dbg_plan[(size_t)unit * 128 + lane] = LA.start[lane >> 4][lane & 15]; A.dbg_plan[(size_t)unit * 128 + 32 + lane] = LA.order[lane >> 4][lane & 15];
        A.dbg_plan[(size_t)unit * 128 + 64 + lane] = LB.start[lane >> 4][lane & 15]; A.dbg_plan[(size_t)unit * 128 + 96 + lane] = LB.order[lane >> 4][lane & 15];
    }
    if (!out.unpaired_out) n_aligned_pairs++;
    else {
        if (out.a.n_best == 1 || (out.a.n_best > 1 && P.report_repeat_hits == 1)) n_aligned++;
        if (out.b.n_best == 1 || (out.b.n_best > 1 && P.report_repeat_hits == 1)) n_aligned++;
    }
    forget_keys(MA, U.SA, lane); forget_keys(MB, U.SB, lane);
}

// one unit in the main kernel; returns true if it was deferred to the heavy pipeline
// (always inlined into the kernel: as a called function its callee-saved registers cost 21 KB of scratch writes per pair,
//  a fifth of the kernel's memory requests — 59.6 ms against 48.6 ms per 2^20 pairs)
// UnitLds: the wave-uniform per-unit state that is not in MateU — slab pointers and the work counters with their value at the
// unit's start (restored when the unit is deferred); in LDS for the same reason (the main kernel is short of scalar registers)
struct UnitLds { UnitSlabs U; Counters C, C0; };
template <bool PE, bool EXACT, bool CTX = false>
__device__ __forceinline__ bool process_unit(const AlignArgs &A, const BlockLds &BL, MateLds &LA, MateLds &LB, uint32_t unit, uint8_t *slab, int lane, UnitLds &UL,
                             u64 &n_aligned, u64 &n_aligned_pairs)
{
    const DevParams &P = A.P;
    const uint32_t hthr = A.heavy_threshold;
    Counters &C = UL.C;
    UL.C0 = C;
    const Counters &C0 = UL.C0;
    UnitSlabs &U = UL.U;
    U = carve_slab(slab, (uint32_t)P.max_snp_num + 1, A.rowcap, PE, A.kcap, A.hbits);
    Mate MA, MB;
    MA.u = lds_mate(&LA.u); MB.u = PE ? lds_mate(&LB.u) : lds_mate(&LA.u2);
    [[clang::always_inline]] unit_prepare<PE, EXACT>(A, BL, LA, LB, MA, MB, unit, lane, C);
    uint32_t pcnt_reg = 0;  // lane c holds _cur_n_hits[c]
    int paired = 0;
    bool defer = false;
    if (PE && !MA.u->filtered && !MB.u->filtered) {
        const int maxi = max(MA.u->max_snp, MB.u->max_snp);  // PairAlign::RunAlign (pairs.cpp:163-172)
        for (int i = 0; i <= maxi && !paired && !defer; i++) {
            if (i < MA.u->seedseg) snp_align<EXACT, PE, CTX>(P, BL, LA, MA, U.SA, i, lane, C, hthr);
            if (!MA.u->defer && i < MB.u->seedseg) snp_align<EXACT, PE, CTX>(P, BL, LB, MB, U.SB, i, lane, C, hthr);
            if (MA.u->defer || MB.u->defer) { defer = true; break; }
            int np_;
            [[clang::always_inline]] np_ = pair_level_post(P, MA, MB, U, pcnt_reg, i, lane);
            if (np_ > 0) paired = i + 1;
        }
    } else {
        if (!MA.u->filtered) { run_align_single<EXACT, PE, CTX>(P, BL, LA, MA, U.SA, lane, C, hthr); defer = MA.u->defer; }
        if (PE && !defer && !MB.u->filtered) { run_align_single<EXACT, PE, CTX>(P, BL, LB, MB, U.SB, lane, C, hthr); defer = MB.u->defer; }
    }
    if (defer) { forget_keys(MA, U.SA, lane); if (PE) forget_keys(MB, U.SB, lane); C = C0; return true; }
    [[clang::always_inline]] unit_finish<PE>(A, LA, LB, MA, MB, U, pcnt_reg, paired, unit, lane, n_aligned, n_aligned_pairs);   // (the main kernel stays a leaf: a call costs it a kilobyte of stack per lane)
    return false;
}

__device__ __forceinline__ void init_block_lds(const DevParams &P, BlockLds &BL, int tid, int nthreads)
{
    for (int i = tid; i < 256; i += nthreads) {
        ((uint8_t *)BL.prof)[i] = ((const uint8_t *)P.profile_a)[i];
        const int k = nt_idx((uint32_t)i);
        BL.nt_tab[i] = (uint8_t)(((P.bit_nt_packed >> (8 * (k < 0 ? 0 : k))) & 3u) | (((P.bit_nt_packed >> (8 * (k < 0 ? 3 : 3 - k))) & 3u) << 2) | (k < 0 ? 0u : 0x30u));
    }
    if (P.n_chr <= BSX_LDS_CHR) {
        for (uint32_t i = tid; i <= P.n_chr; i += nthreads) BL.anchor[i] = P.anchor[i];
        for (uint32_t i = tid; i < P.n_chr; i += nthreads) { BL.chr_size[i] = P.chr_size[i]; BL.rc_offset[i] = P.rc_offset[i]; }
    }
}

__device__ __forceinline__ void flush_counters(const AlignArgs &A, const Counters &C, u64 n_units_done, u64 n_aligned, u64 n_aligned_pairs, bool main_kernel = false)
{
    atomicAdd((u64 *)&A.counters[0], C.n_lookup); atomicAdd((u64 *)&A.counters[1], C.n_cand);
    atomicAdd((u64 *)&A.counters[2], C.sum_w); atomicAdd((u64 *)&A.counters[3], C.n_orient);
    if (main_kernel) {  // the main kernel's own share (counters 11-14)
        atomicAdd((u64 *)&A.counters[11], C.n_lookup); atomicAdd((u64 *)&A.counters[12], C.n_cand);
        atomicAdd((u64 *)&A.counters[13], C.sum_w); atomicAdd((u64 *)&A.counters[14], C.n_orient);
    }
    atomicAdd((u64 *)&A.counters[4], n_units_done); atomicAdd((u64 *)&A.counters[5], n_aligned);
    atomicAdd((u64 *)&A.counters[6], n_aligned_pairs);
}

#ifndef BSX_WAVES_PER_EU_SE
#define BSX_WAVES_PER_EU_SE 6
#endif
#ifndef BSX_WAVES_PER_EU_PE
#define BSX_WAVES_PER_EU_PE 5  /* round 4: 96 VGPRs, 112 B of scratch per lane; five waves per SIMD of a kernel that waits on memory 76 % of its cycles: 112.4-113.4 against 114.6-114.7 ms per step (4 waves: 128 VGPRs, 56 B) */
#endif
// main kernel: persistent waves, one unit per wave at a time
template <bool PE, bool EXACT, bool CTX = false>
__global__ __launch_bounds__(256, PE ? BSX_WAVES_PER_EU_PE : BSX_WAVES_PER_EU_SE) void k_align(AlignArgs A)
{
    __shared__ BlockLds BL;
    __shared__ WaveLds<PE> WL[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    init_block_lds(A.P, BL, threadIdx.x, 256);
    __syncthreads();
    const uint32_t slot = blockIdx.x * 4 + wv;
    MateLds &LA = WL[wv].mate[0];
    MateLds &LB = WL[wv].mate[PE ? 1 : 0];
    __shared__ UnitLds ULS[4];
    UnitLds &UL = ULS[wv];
    Counters &C = UL.C;
    C.n_lookup = 0; C.n_cand = 0; C.sum_w = 0; C.n_orient = 0;
    u64 n_units_done = 0, n_aligned = 0, n_aligned_pairs = 0;
    for (;;) {
        uint32_t unit = 0;
        if (lane == 0) unit = atomicAdd(A.queue, 1u);
        unit = rfl(unit) + A.first_unit;
        if (unit >= A.n_units) break;
        if (A.unit_list) unit = rfl(A.unit_list[unit]);  // redo run: the units named by the list
        const u64 t_begin = A.dbg_cycles ? __builtin_readcyclecounter() : 0;
        uint8_t *slab = A.scratch + (size_t)(A.debug ? unit : slot) * A.slab_bytes;
        const bool deferred = process_unit<PE, EXACT, CTX>(A, BL, LA, LB, unit, slab, lane, UL, n_aligned, n_aligned_pairs);
        if (deferred) { if (lane == 0) A.heavy_list[atomicAdd(A.heavy_count, 1u)] = unit; }
        else n_units_done++;
        if (A.dbg_cycles && lane == 0) A.dbg_cycles[unit] = (uint32_t)min((u64)0xffffffffull, (u64)__builtin_readcyclecounter() - t_begin);
        wave_fence();
    }
    if (lane == 0) flush_counters(A, C, n_units_done, n_aligned, n_aligned_pairs, true);
}

// ---------------------------------------------------------------------------------------------------------------
// heavy pipeline: units whose candidate lists are too long for one wave
// ---------------------------------------------------------------------------------------------------------------
// A bucket of a low-complexity seed holds up to millions of entries and the reference walks it candidate by candidate.
// Deferred units advance in lock-step iterations of two kernels:
//   k_hctrl : one wave per unit runs the unit's control flow (exactly the main kernel's logic) until it needs a long
//             list scanned; it then publishes a window of that list as tasks of HS_TASK candidates and saves its state
//   k_hscan : every wave of the chip pulls tasks, evaluates their candidates and leaves an ordered survivor record
// The owner replays the survivors in order on the next k_hctrl pass.  An event that lowers the threshold or ends the
// call cuts the window right after the candidate that caused it, so every candidate is still evaluated under exactly
// the threshold the reference would have used: results and work counters are bit-identical to the one-wave path.
// Windows grow geometrically so that an early exit wastes at most a bounded amount of scanning.
#ifndef HS_TASK
#define HS_TASK 8192u
#endif
#ifndef HS_TASK_MIN
#define HS_TASK_MIN 1024u  // inside the heavy pipeline even moderately long lists go to the scan kernel
#endif
#ifndef HS_SCAP
#define HS_SCAP 512u
#endif
#ifndef HS_WIN0
#define HS_WIN0 1048576u  // early exits inside long lists are rare and a scanned candidate costs ~4 ps of chip time
#endif
#ifndef HS_GROW
#define HS_GROW 4u
#endif
#ifndef BSX_EVENT_CONTINUE
#define BSX_EVENT_CONTINUE 1  /* work counters off: a lowered threshold does not end a window's replay (snp_align_heavy) */
#endif
#ifndef HS_WINMAX
#define HS_WINMAX (1u << 22)
#endif
#ifndef HS_WINTAIL
#define HS_WINTAIL (1u << 27)  /* window of a list once fewer than 256 units of the round are active (through round 5: HS_WINMAX) */
#endif
#ifndef HS_TAILU_A
#define HS_TAILU_A 256u    /* active units of the round below which a list goes out whole (HS_WINTAIL) */
#define HS_TAILU_B 2048u   /* ... below which windows are HS_WINTAIL2K */
#endif
#ifndef HS_WINTAIL2K
#define HS_WINTAIL2K (1u << 24)  /* ... fewer than 2048 (2048 tasks per list: 2048 such units can fill a 2 M-task pool twice) */
#endif

struct SurvRec { uint32_t w_ord, hchr, hloc, hkey; };  // w in bits 0-7, ordinal inside the task in bits 8+

struct ListReq { uint32_t nsub, total, nwords, len, thres, rrbs, tag_xor, tag_want;  // rrbs: the list is one bucket of {tag, loc} pairs; entries with ((tag ^ tag_xor) >> 16) == tag_want are its candidates
                 uint32_t sub_pre[32], sub_n[32], sub_base[32], sub_h[32]; uint32_t rw[12], rm[12];
                 uint32_t px[8], py[8], pm[8]; };  // the read as bit planes (bsx_dev.h): low bits, high bits, not-N bits of nt [32 j, 32 j + 32); words 5-7 zero
struct HMate {
    int32_t len, raw_len, max_snp, seedseg, filtered;
    uint32_t flags, snp_thres, nkeys, index, nfull, pad[6];
    uint32_t cnt_reg[64], key_reg[64], bloom0[64], bloom1[64];
    uint32_t w[2][10], m[2][10];
    uint8_t start[2][16], order[2][16];
    uint32_t stale_key[2][16];  // "-p 1 exact" mode: make_list needs the tail entries again on later visits
};
#define HS_NSLOT 4  /* list slots of a unit: slot = 2 * mate + read orientation — the list each mate is working on, and (round 6, work counters off) the other orientation's list of the same SnpAlign call, published ahead */
#ifndef BSX_SPECULATE
#define BSX_SPECULATE 1  /* snp_align_heavy: publish the complementary-chain list together with the last window of the direct-chain list */
#endif
struct HState {
    uint32_t want;  // tasks of a request the pool refused (0: none): the unit is only restored once that many are free
    int32_t level, sub;
    int32_t paired;
    // per list cursor (slot = mate): both mates of a pair publish their lists of a level in the same visit (their SnpAlign calls
    // are independent, pairs.cpp:165-166), so each keeps its own orientation / position / window
    int32_t orient[2], have[HS_NSLOT];
    uint32_t c[2], W[2];
    uint32_t t0[HS_NSLOT], n_tasks[HS_NSLOT], win_c0[HS_NSLOT], win_n[HS_NSLOT];  // the published window: tasks t0.. cover candidates [win_c0, win_c0+win_n)
    Counters C;
    uint32_t pcnt_reg[64];
    HMate mate[2];
    ListReq req[HS_NSLOT];
};
struct HTask { uint32_t h, c0, n, key;   // h: unit (bits 0-29) and list slot (bits 30-31); key: index entry the task starts at (tasks are scanned in key order, see bsx_api.hip)
               uint32_t sub_h, flags, pad[2]; };  // of the sub-range the task starts in: its offset h; flags bit 0 strand copy, bit 1 the task lies inside that one sub-range, bit 2 RRBS list, bits 8-11 the read's 32-nt words; pad: RRBS tag filter
struct HTaskOut { uint32_t count, overflow, acc[4], c0, n; SurvRec surv[HS_SCAP]; };  // c0, n: the task's candidates [c0, c0 + n) of its list, echoed by the scan kernel (the replay reads the geometry of a window from here: task descriptors are reused during a control pass, outputs are not)
struct HeavyArgs {
    HState *state; uint8_t *slabs; uint32_t *active_in, *active_out, *n_active_out; HTask *tasks; HTaskOut *tout; uint32_t *n_tasks, *queue;
    const uint32_t *n_active_in_ptr;
    uint32_t n_active_in, task_cap, fresh, list_base, hidx_base;
    const uint32_t *order;  // scan order of the tasks (task ids sorted by key), null = pool order
    uint32_t xcd_map;       // 1: blocks of one XCD take a contiguous part of the order
    const uint32_t *ghead, *glist;  // k_hscan_same: size of the group that starts at a scan slot; the start slots, their count in glist[task_cap] (k_task_groups)
};
__host__ HeavyArgs typed(const HeavyArgsRaw &r)
{
    HeavyArgs h;
    h.state = (HState *)r.state; h.slabs = r.slabs; h.active_in = r.active_in; h.active_out = r.active_out; h.n_active_out = r.n_active_out;
    h.tasks = (HTask *)r.tasks; h.tout = (HTaskOut *)r.tout; h.n_tasks = r.n_tasks; h.queue = r.queue;
    h.n_active_in_ptr = r.n_active_in_ptr; h.n_active_in = r.n_active_in; h.task_cap = r.task_cap; h.fresh = r.fresh; h.list_base = r.list_base; h.hidx_base = r.hidx_base;
    h.order = r.order; h.xcd_map = r.xcd_map; h.ghead = r.ghead; h.glist = r.glist;
    return h;
}

__device__ void save_mate(HMate &d, const Mate &M, const MateLds &L, int lane)
{
    if (lane == 0) {
        d.len = M.u->len; d.raw_len = M.u->raw_len; d.max_snp = M.u->max_snp; d.seedseg = M.u->seedseg; d.filtered = M.u->filtered;
        d.flags = M.u->flags; d.snp_thres = M.u->snp_thres; d.nkeys = M.u->nkeys; d.index = M.u->index; d.nfull = (uint32_t)M.u->nfull;
    }
    d.cnt_reg[lane] = M.cnt_reg; d.key_reg[lane] = M.key_reg; d.bloom0[lane] = M.bloom0; d.bloom1[lane] = M.bloom1;
    if (lane < 20) { (&d.w[0][0])[lane] = (&L.w[0][0])[lane]; (&d.m[0][0])[lane] = (&L.m[0][0])[lane]; }
    if (lane < 32) { (&d.start[0][0])[lane] = (&L.start[0][0])[lane]; (&d.order[0][0])[lane] = (&L.order[0][0])[lane]; (&d.stale_key[0][0])[lane] = (&L.stale_key[0][0])[lane]; }
}

__device__ void load_mate(const HMate &d, Mate &M, MateLds &L, int lane)
{
    M.u->len = (int)rfl((uint32_t)d.len); M.u->raw_len = (int)rfl((uint32_t)d.raw_len); M.u->max_snp = (int)rfl((uint32_t)d.max_snp);
    M.u->seedseg = (int)rfl((uint32_t)d.seedseg); M.u->filtered = (int)rfl((uint32_t)d.filtered);
    M.u->flags = rfl(d.flags); M.u->snp_thres = rfl(d.snp_thres); M.u->nkeys = rfl(d.nkeys); M.u->index = rfl(d.index); M.u->nfull = (int)rfl(d.nfull);
    M.u->defer = 0;
    M.cnt_reg = d.cnt_reg[lane]; M.key_reg = d.key_reg[lane]; M.bloom0 = d.bloom0[lane]; M.bloom1 = d.bloom1[lane];
    if (lane < 20) { (&L.w[0][0])[lane] = (&d.w[0][0])[lane]; (&L.m[0][0])[lane] = (&d.m[0][0])[lane]; }
    if (lane < 32) { (&L.start[0][0])[lane] = (&d.start[0][0])[lane]; (&L.order[0][0])[lane] = (&d.order[0][0])[lane]; (&L.stale_key[0][0])[lane] = (&d.stale_key[0][0])[lane]; }
    wave_fence();
}

struct HCursor { int level, sub, paired; int orient[2], have[HS_NSLOT]; uint32_t c[2], W[2], n_active, want; u64 vc[8]; uint32_t vn[8]; };  // vc/vn: per-visit category clocks and counts (diagnostics)

// diagnostic category clocks of k_hctrl (only when the caller asked for unit cycles): 0 prepare/restore, 1 inline scans,
// 2 survivor replay, 3 sort+pairs, 4 save/finish, 5 recount after events, (6 the whole advance,) 7 the rest of an overflowed task behind its recorded survivors
#define CAT_BEGIN(A) const u64 cat_t0_ = (A).dbg_cat ? __builtin_readcyclecounter() : 0
#define CAT_END(A, k) do { if ((A).dbg_cat) { const u64 d_ = (u64)__builtin_readcyclecounter() - cat_t0_; K.vc[k] += d_; K.vn[k]++; if (lane == 0) { atomicAdd((u64 *)&(A).dbg_cat[k], d_); atomicMax((u64 *)&(A).dbg_cat[8 + (k)], d_); } } } while (0)

// a survivor record as k_hscan leaves it (strand copy in hchr, global position in hloc) -> hit coordinates in place;
// returns the lanes of `act` whose candidate lies inside its chromosome (align.cpp:273)
__device__ __forceinline__ u64 surv_coords(const DevParams &P, const BlockLds &BL, SurvRec &r, int len, int lane, u64 act)
{
    bool ok = (act >> lane) & 1;
    if (ok) {
        uint32_t hchr, hloc, hkey;
        ok = hit_coords(P, BL, r.hloc, r.hchr, len, hchr, hloc, hkey);
        r.hchr = hchr; r.hloc = hloc; r.hkey = hkey;
    }
    return bsx_ballot(ok);
}

// How a window of a list is cut into scan tasks.  Thousands of reads walk the same giant bucket in the same pass; cut at multiples
// of HS_TASK from each read's own window start, their tasks would cover the bucket's entries in shifted pieces.  Instead every
// LARGE sub-range (>= HS_TASK candidates: one strand part of one bucket) is cut on the grid of ABSOLUTE index entries (entry
// index mod HS_TASK == 0), so that the tasks of all the reads that walk it cover identical entry ranges — which is what lets the
// scan kernel evaluate many reads against one fetch of the candidates' reference windows (k_hscan_multi).  A task never leaves
// its segment: a large sub-range, or a maximal run of small ones (cut every HS_TASK from the run's start, as before).
// Lane i < n holds segment i: candidates [lo, hi) of the list, cut points at ordinals c with (c - ph) % HS_TASK == 0.
struct SegTab { uint32_t lo, hi, ph; int n; };
__device__ __forceinline__ SegTab list_segments(const CandList &cl, int lane)
{
    SegTab g; g.lo = g.hi = g.ph = 0;
    int n = 0;
    bool open = false;  // the current segment is a run of small sub-ranges
    for (int s_ = 0; s_ < cl.nsub; s_++) {
        const uint32_t ps = rl(cl.sub_pre, s_), ns = rl(cl.sub_n, s_), sb = rl(cl.sub_base, s_);
        if (ns == 0) continue;
        if (ns >= HS_TASK) {
            if (lane == n) { g.lo = ps; g.hi = ps + ns; g.ph = ps - (sb % HS_TASK); }  // entry index of ordinal c is sb + (c - ps)  (mod 2^32 throughout: HS_TASK divides it)
            n++; open = false;
        } else if (open) { if (lane == n - 1) g.hi = ps + ns; }
        else { if (lane == n) { g.lo = ps; g.hi = ps + ns; g.ph = ps; } n++; open = true; }
    }
    g.n = n;
    return g;
}
// tasks of window [c0, c0 + wn): lane i gets the number of tasks its segment contributes (cnt), their first cell (cell0) and the part of the window inside the segment
struct SegCut { uint32_t cnt, cell0, lo, hi; };
__device__ __forceinline__ SegCut window_cuts(const SegTab &g, uint32_t c0, uint32_t wn, int lane)
{
    SegCut w; w.cnt = 0; w.cell0 = 0;
    w.lo = max(c0, g.lo); w.hi = min(c0 + wn, g.hi);
    if (lane < g.n && w.lo < w.hi) { w.cell0 = (w.lo - g.ph) / HS_TASK; w.cnt = (w.hi - 1u - g.ph) / HS_TASK - w.cell0 + 1u; }
    return w;
}

// publish candidates [c0, c0 + wn) of list `cl` as scan tasks of list slot `slot`; false if the task pool cannot take them
__device__ __forceinline__ bool publish_window(const DevParams &P, const HeavyArgs &H, HState *S, uint32_t hidx, int slot, const CandList &cl, int orient, int seg,
                                               const MateLds &L, const Mate &M, uint32_t c0, uint32_t wn, int lane, uint32_t &nt_out)
{
    const SegTab G = list_segments(cl, lane);
    const SegCut W = window_cuts(G, c0, wn, lane);
    uint32_t pre = W.cnt;  // inclusive prefix over segments
#pragma unroll
    for (int o_ = 1; o_ < 32; o_ <<= 1) { const uint32_t v_ = __shfl_up(pre, o_); if (lane >= o_) pre += v_; }
    const uint32_t nt = rl(pre, 31);   // (at most 32 segments)
    nt_out = nt;
    uint32_t t0 = 0;
    if (lane == 0) t0 = atomicAdd(H.n_tasks, nt);
    t0 = rfl(t0);
    if (t0 + nt > H.task_cap) {
        if (t0 < H.task_cap)  // pool exhausted mid-way: neutralise the slots that were reserved
            for (uint32_t t = t0 + lane; t < H.task_cap; t += 64) { HTask tk; tk.h = hidx; tk.c0 = 0; tk.n = 0; tk.key = 0xffffffffu; tk.sub_h = 0; tk.flags = 0; tk.pad[0] = tk.pad[1] = 0; H.tasks[t] = tk; }
        return false;
    }
    for (uint32_t tb = 0; tb < nt; tb += 64) {
        const uint32_t t = tb + lane;
        HTask tk; tk.h = hidx | ((uint32_t)slot << 30); tk.c0 = 0; tk.n = 0; tk.key = 0;
        for (int i_ = 0; i_ < G.n; i_++) {  // the segment task t lies in
            const uint32_t end_ = rl(pre, i_), cnt_ = rl(W.cnt, i_);
            if (cnt_ && t >= end_ - cnt_ && t < end_) {
                const uint32_t j_ = t - (end_ - cnt_), cell = rl(W.cell0, i_) + j_, ph = rl(G.ph, i_);
                // (ph + cell * HS_TASK may lie before ordinal 0 for the first cell — a wrapped number: only cells behind the first start on the grid)
                const uint32_t a_ = j_ ? ph + cell * HS_TASK : rl(W.lo, i_), b_ = min(rl(W.hi, i_), ph + (cell + 1u) * HS_TASK);
                tk.c0 = a_; tk.n = b_ - a_;
            }
        }
        tk.sub_h = 0; tk.flags = 0;
        tk.pad[0] = P.rrbs && orient ? 0x1000000u : 0u; tk.pad[1] = !P.rrbs ? 0u : orient ? (uint32_t)(M.u->nfull - 1 - seg) : (uint32_t)seg;   // RRBS: the list's tag filter (ListReq::tag_xor, tag_want)
        for (int s_ = 0; s_ < cl.nsub; s_++) {  // the index entry the task starts at, and what a scan kernel needs to know about that sub-range
            const uint32_t ps_ = rl(cl.sub_pre, s_), ns_ = rl(cl.sub_n, s_), sb_ = rl(cl.sub_base, s_), sh_ = rl(cl.sub_h, s_);
            if (tk.c0 >= ps_ && tk.c0 < ps_ + ns_) { tk.key = sb_ + (tk.c0 - ps_); tk.sub_h = sh_; tk.flags = ((uint32_t)s_ & 1u) | (tk.c0 + tk.n <= ps_ + ns_ ? 2u : 0u) | (P.rrbs ? 4u : 0u) | ((uint32_t)((M.u->len + 31) >> 5) << 8); }
        }
        if (t < nt) H.tasks[t0 + t] = tk;
    }
    ListReq &R = S->req[slot];
    if (lane < 32) { R.sub_pre[lane] = cl.sub_pre; R.sub_n[lane] = cl.sub_n; R.sub_base[lane] = cl.sub_base; R.sub_h[lane] = cl.sub_h; }
    if (lane < 9) { R.rw[lane] = L.w[orient][lane]; R.rm[lane] = L.m[orient][lane]; }
    if (lane < 8) {  // the same read as bit planes for the scan kernels (L.w / L.m hold 10 packed words; a not-N nt has both mask bits set)
        const bool in = lane < 5;
        const uint32_t wa = in ? L.w[orient][2 * lane] : 0u, wb = in ? L.w[orient][2 * lane + 1] : 0u, ma = in ? L.m[orient][2 * lane] : 0u, mb = in ? L.m[orient][2 * lane + 1] : 0u;
        R.px[lane] = bsx_plane_word(wa, wb, 0); R.py[lane] = bsx_plane_word(wa, wb, 1); R.pm[lane] = bsx_plane_word(ma, mb, 0);
    }
    if (lane == 0) {
        R.nsub = (uint32_t)cl.nsub; R.total = cl.total; R.nwords = (uint32_t)((M.u->len + 15) >> 4); R.len = (uint32_t)M.u->len; R.thres = M.u->snp_thres;
        R.rrbs = P.rrbs ? 1u : 0u;  // tag filter of align.cpp:187,229: forward reads want their segment, rc reads cmodeindex with the direction bit flipped
        R.tag_xor = orient ? 0x1000000u : 0u; R.tag_want = orient ? (uint32_t)(M.u->nfull - 1 - seg) : (uint32_t)seg;
        S->t0[slot] = t0; S->n_tasks[slot] = nt; S->win_c0[slot] = c0; S->win_n[slot] = wn;
    }
    return true;
}

// resumable SnpAlign for a deferred unit: 0 = call complete, 1 = the reference's SnpAlign returned early, 2 = a window
// of the current list was published and the unit must wait for k_hscan
// SPEC: the instantiation carries the publish-ahead code (paired batches only: in the single-end control kernel the extra code lifted the register count from 224
// to 256 — a control wave then leaves room for two scan waves beside it on its SIMD instead of three, and C4 with three batches in flight went from 415 to
// 464 ms per step although the code never ran there, gpurun_out/r06f, r06h)
template <bool SPEC>
__device__ __forceinline__ int snp_align_heavy(const AlignArgs &A, const HeavyArgs &H, HState *S, uint32_t hidx, const BlockLds &BL, const MateLds &L, Mate &M,
                               const Slab &SL, int mode, HCursor &K, int ms, int lane, Counters &C)
{
    const DevParams &P = A.P;
    for (; K.orient[ms] < 2; K.orient[ms]++, K.c[ms] = 0, K.W[ms] = HS_WIN0) {
        const int orient = K.orient[ms];
        const int slot = 2 * ms + orient;
        if (!((M.u->flags >> orient) & 1)) continue;
        const int seg = L.order[orient][mode];
        const CandList cl = make_list<true>(P, BL, L, M, orient, seg, lane);
        if (cl.total < min(A.heavy_threshold, (uint32_t)HS_TASK_MIN)) {  // short list: the owning wave scans it itself
            CAT_BEGIN(A);
            const int r_ = wave_scan_range<false, 4>(P, BL, L, M, SL, cl, orient, seg, mode, 0, cl.total, 0, lane, C);
            CAT_END(A, 1);
            if (r_ == 2) { wave_fence(); return 1; }
            continue;
        }
        while (K.c[ms] < cl.total) {
            if (K.have[slot]) {
                K.have[slot] = 0;
                // (task descriptors in the pool may already have been reused by other units of this pass: the window is
                //  reconstructed from the unit's own state, only the task OUTPUTS are read from the pool)
                const uint32_t t0 = rfl(S->t0[slot]), nt = rfl(S->n_tasks[slot]), req_thres = rfl(S->req[slot].thres);
                const bool cont_events = BSX_EVENT_CONTINUE && !A.work_counters;
                bool restart = false;
                for (uint32_t tg = 0; tg < nt && !restart; tg += 64) {
                    // 64 task headers at a time: tasks without survivors only contribute their work counters
                    const uint32_t tl = tg + lane;
                    uint32_t hc = 0, hov = 0, h0 = 0, hw = 0, gc0 = 0, gn_ = 0;  // gc0 / gn_: the task's candidates [gc0, gc0 + gn_) (publish_window's cuts)
                    if (tl < nt) {
                        const HTaskOut *oh = &H.tout[t0 + tl];
                        hc = oh->count; hov = oh->overflow; h0 = oh->acc[0]; hw = oh->acc[1] + 2 * oh->acc[2] + 5 * oh->acc[3]; gc0 = oh->c0; gn_ = oh->n;
                    }
                    u64 special = bsx_ballot(tl < nt && (hc != 0 || hov != 0));
                    uint32_t done_upto = 0;  // tasks [tg, tg+done_upto) of this group are fully accounted
                    const uint32_t gn = min(64u, nt - tg);
                    while (!restart) {
                        {
                            // consecutive tasks from done_upto whose survivor records fit one 64-lane group (none overflowed)
                            // are replayed together: one round of record loads and hitset probes instead of one per task
                            const bool inr = (uint32_t)lane >= done_upto && (uint32_t)lane < gn;
                            const u64 ovm = bsx_ballot(inr && hov != 0);
                            const uint32_t first_ov = ovm ? (uint32_t)__builtin_ctzll(ovm) : gn;
                            uint32_t ps = inr ? hc : 0;  // inclusive prefix sum of the survivor counts
                            for (int o_ = 1; o_ < 64; o_ <<= 1) { const uint32_t v_ = __shfl_up(ps, o_); if (lane >= o_) ps += v_; }
                            const u64 fit = bsx_ballot(inr && (uint32_t)lane < first_ov && ps <= 64);
                            const uint32_t bend = done_upto + (uint32_t)__builtin_popcountll(fit);
                            if (bend > done_upto) {
                                const uint32_t total = rl(ps, (int)bend - 1);
                                int e = 0, ls = -1;
                                uint32_t my_t = 0;
                                SurvRec r = {0, 0, 0, 0};
                                if (total) {
                                    CAT_BEGIN(A);
                                    uint32_t my_i = 0;
                                    for (u64 sm = bsx_ballot(inr && (uint32_t)lane < bend && hc != 0); sm; sm &= sm - 1) {
                                        const int t_ = (int)__builtin_ctzll(sm);
                                        const uint32_t end_ = rl(ps, t_), beg_ = end_ - rl(hc, t_);
                                        if ((uint32_t)lane >= beg_ && (uint32_t)lane < end_) { my_t = (uint32_t)t_; my_i = (uint32_t)lane - beg_; }
                                    }
                                    if ((uint32_t)lane < total) r = H.tout[t0 + tg + my_t].surv[my_i];
                                    u64 m = total >= 64 ? ~0ull : ((1ull << total) - 1);
                                    m &= surv_coords(P, BL, r, M.u->len, lane, m);
                                    for (;;) {
                                        if (__builtin_popcountll(m) > BSX_GROUP_MIN) e = accept_group(P, M, SL, orient, mode, m, r.w_ord & 0xff, r.hchr, r.hloc, r.hkey, lane, ls);
                                        else {
                                            e = 0;
                                            while (m) {
                                                const int l1 = (int)__builtin_ctzll(m);
                                                m &= m - 1;
                                                e = accept_survivor<true>(P, M, SL, orient, mode, rl(r.w_ord, l1) & 0xff, rl(r.hchr, l1), rl(r.hloc, l1), rl(r.hkey, l1), lane);
                                                if (e) { ls = l1; break; }
                                            }
                                        }
                                        // A lowered threshold (align.cpp:278) ends the window only where the work counters are kept (the early-out classes of the
                                        // candidates behind it depend on it).  The HITS do not: a survivor's record carries its exact count, and accept_* tests it
                                        // against the CURRENT threshold — the survivors behind the event are simply replayed under the new one.
                                        if (e != 1 || !cont_events) break;
                                        m &= ls >= 63 ? 0ull : ~((2ull << ls) - 1ull);
                                        e = 0;
                                        if (!m) break;
                                    }
                                    CAT_END(A, 2);
                                }
                                const uint32_t upto = e ? rl(my_t, ls) : bend;  // tasks [done_upto, upto) are complete
                                const bool mine_b = (uint32_t)lane >= done_upto && (uint32_t)lane < upto;
                                C.n_cand += wave_sum(mine_b ? h0 : 0);
                                C.sum_w += wave_sum(mine_b ? hw : 0);
                                if (!e) {
                                    done_upto = bend;
                                    special &= bend >= 64 ? 0ull : ~((1ull << bend) - 1);
                                    if (bend >= gn) { K.c[ms] = rl(gc0, (int)gn - 1) + rl(gn_, (int)gn - 1); break; }
                                    continue;
                                }
                                // count exactly the candidates of the event's task up to and including the one that caused it
                                const uint32_t tc0e = rl(gc0, (int)upto), Xe = tc0e + (rl(r.w_ord, ls) >> 8);
                                if (A.work_counters) { CAT_BEGIN(A); wave_scan_range<true, 4>(P, BL, L, M, SL, cl, orient, seg, mode, tc0e, Xe + 1, req_thres, lane, C); CAT_END(A, 5); }   // (a count-only walk: nothing but the work counters depends on it)
                                K.c[ms] = Xe + 1;
                                if (e == 2) { wave_fence(); return 1; }
                                restart = true;
                                continue;
                            }
                        }
                        const uint32_t nxt = special ? (uint32_t)__builtin_ctzll(special) : gn;  // next task needing a replay (overflowed, or more than 64 survivors)
                        // plain tasks in [done_upto, nxt)
                        const bool mine = (uint32_t)lane >= done_upto && (uint32_t)lane < nxt;
                        C.n_cand += wave_sum(mine ? h0 : 0);
                        C.sum_w += wave_sum(mine ? hw : 0);
                        if (nxt >= gn) { K.c[ms] = rl(gc0, (int)gn - 1) + rl(gn_, (int)gn - 1); break; }
                        special &= special - 1;
                        done_upto = nxt + 1;
                        const uint32_t t = tg + nxt;
                        const uint32_t tc0 = rl(gc0, (int)nxt), tn = rl(gn_, (int)nxt);
                        const HTaskOut *o = &H.tout[t0 + t];
                        // A task with more survivors than its record holds (HS_SCAP) carries the FIRST HS_SCAP of them, in list order.  With the work counters on the
                        // owning wave redoes the whole task itself (exact counters).  Without them the recorded prefix is replayed like any other record, and only
                        // what lies behind its last survivor is still open (below) — a one-wave scan of 8 192 candidates is 128 dependent round trips, and
                        // short reads in repeats overflow in most tasks of their first windows (C5).
                        const bool ovt = rl(hov, (int)nxt) != 0;
                        if (ovt && !cont_events) {  // too many survivors for the record: redo this task with the one-wave path
                            CAT_BEGIN(A);
                            const int r = wave_scan_range<false, 4>(P, BL, L, M, SL, cl, orient, seg, mode, tc0, tc0 + tn, 0, lane, C);
                            CAT_END(A, 1);
                            if (r == 2) { wave_fence(); return 1; }
                            K.c[ms] = tc0 + tn;
                            if (r == 1 && !cont_events) restart = true;  // later tasks were evaluated under the old threshold (their work counters, not their hits, depend on it)
                            continue;
                        }
                        int event = 0; uint32_t X = 0;
                        const uint32_t nv = ovt ? (uint32_t)HS_SCAP : rl(hc, (int)nxt);
                        CAT_BEGIN(A);
                        for (uint32_t base = 0; base < nv && !event; base += 64) {
                            const uint32_t i = base + lane;
                            SurvRec r = {0, 0, 0, 0};
                            if (i < nv) r = o->surv[i];
                            u64 m = bsx_ballot(i < nv);
                            m &= surv_coords(P, BL, r, M.u->len, lane, m);
                            while (__builtin_popcountll(m) > BSX_GROUP_MIN) {
                                int ls;
                                const int e = accept_group(P, M, SL, orient, mode, m, r.w_ord & 0xff, r.hchr, r.hloc, r.hkey, lane, ls);
                                if (e == 1 && cont_events) { m &= ls >= 63 ? 0ull : ~((2ull << ls) - 1ull); continue; }  // (see above: the rest under the new threshold)
                                if (e) { event = e; X = tc0 + (rl(r.w_ord, ls) >> 8); }
                                m = 0;
                            }
                            while (m) {
                                const int ls = (int)__builtin_ctzll(m);
                                m &= m - 1;
                                const uint32_t wo = rl(r.w_ord, ls);
                                const int e = accept_survivor<true>(P, M, SL, orient, mode, wo & 0xff, rl(r.hchr, ls), rl(r.hloc, ls), rl(r.hkey, ls), lane);
                                if (e == 1 && cont_events) continue;
                                if (e) { event = e; X = tc0 + (wo >> 8); break; }
                            }
                        }
                        CAT_END(A, 2);
                        if (!event && ovt) {
                            // (work counters off) behind the last recorded survivor: if the threshold has fallen since the window was published, the scan kernel
                            // will find fewer survivors there — the window ends here and the rest is published again (at most max_snp such cuts per list);
                            // otherwise the rest of this one task is walked by the wave itself, as the whole task used to be
                            const uint32_t c_next = tc0 + (rfl(o->surv[HS_SCAP - 1u].w_ord) >> 8) + 1u;
                            if (M.u->snp_thres < req_thres) { K.c[ms] = c_next; restart = true; }
                            else {
                                CAT_BEGIN(A);
                                const int r = c_next < tc0 + tn ? wave_scan_range<false, 4>(P, BL, L, M, SL, cl, orient, seg, mode, c_next, tc0 + tn, 0, lane, C) : 0;
                                CAT_END(A, 7);   // (its own clock: what is left of the one-wave scans of overflowed tasks)
                                if (r == 2) { wave_fence(); return 1; }
                                K.c[ms] = tc0 + tn;
                            }
                        } else if (!event) {
                            C.n_cand += rl(h0, (int)nxt);
                            C.sum_w += rl(hw, (int)nxt);
                            K.c[ms] = tc0 + tn;
                        } else {  // count exactly the candidates up to and including the one that caused the event
                            if (A.work_counters) { CAT_BEGIN(A); wave_scan_range<true, 4>(P, BL, L, M, SL, cl, orient, seg, mode, tc0, X + 1, req_thres, lane, C); CAT_END(A, 5); }
                            K.c[ms] = X + 1;
                            if (event == 2) { wave_fence(); return 1; }
                            restart = true;
                        }
                    }
                }
                if (!restart) K.W[ms] = min(K.W[ms] * HS_GROW, (uint32_t)HS_WINMAX);
            } else {
                // few units left: the scan kernel's capacity is idle and every further window of a list is one more pass of the batch's tail (a control visit, three
                // order kernels, a scan launch) — the whole list goes out at once (a microsatellite bucket is 3.9 M entries per strand part, a list 8-30 M candidates)
                const uint32_t weff = K.n_active < HS_TAILU_A ? (uint32_t)HS_WINTAIL : K.n_active < HS_TAILU_B ? (uint32_t)HS_WINTAIL2K : K.W[ms];
                const uint32_t tfit = H.task_cap > 128u ? H.task_cap - 64u : H.task_cap / 2u;   // (a window of w candidates makes at most w / HS_TASK + 2 tasks per segment, 32 segments)
                const uint32_t wpool = (u64)tfit * HS_TASK < (u64)weff ? tfit * HS_TASK : weff;  // a window must fit the task pool
                const uint32_t wn = min(wpool, cl.total - K.c[ms]);
                uint32_t nt = 0;
                if (orient == 0) K.have[slot + 1] = 0;   // (a list published ahead is only good for the very next pass: its records live one pass)
                if (publish_window(P, H, S, hidx, slot, cl, orient, seg, L, M, K.c[ms], wn, lane, nt)) {
                    K.have[slot] = 1;
                    // Publish ahead (work counters off).  This window takes the direct-chain list to its end: the next thing the call does is the complementary-chain
                    // list (align.cpp:300), under a threshold that can only have fallen by then — its first window goes out now, under today's threshold, and the
                    // replay takes from its records what the threshold of the moment admits (as behind any lowered threshold).  One control pass per SnpAlign call and
                    // mate instead of two.  If the call returns inside the direct chain (align.cpp:277) the records are never read.  Only while the task pool is
                    // less than half full: a refused request of ANOTHER unit costs that unit a pass.
                    // Not for RRBS: its rounds fill the task pool (1.26 M tasks per 125 K units against 1.4 M), and lists published ahead push other units'
                    // requests out of the pass — C4 464 against 415 ms per step with three batches in flight (gpurun_out/r06f).
                    if (BSX_SPECULATE && SPEC && !A.work_counters && !P.rrbs && orient == 0 && K.c[ms] + wn == cl.total && ((M.u->flags >> 1) & 1)) {
                        const int seg1 = L.order[1][mode];
                        const CandList cl1 = make_list<true>(P, BL, L, M, 1, seg1, lane);
                        const uint32_t w1 = min(K.n_active < HS_TAILU_A ? (uint32_t)HS_WINTAIL : K.n_active < HS_TAILU_B ? (uint32_t)HS_WINTAIL2K : (uint32_t)HS_WIN0, min(wpool, cl1.total));
                        if (cl1.total >= min(A.heavy_threshold, (uint32_t)HS_TASK_MIN) &&
                            rfl(__hip_atomic_load(H.n_tasks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) + w1 / HS_TASK + 66u <= H.task_cap / 2u) {
                            uint32_t nt1 = 0;
                            if (publish_window(P, H, S, hidx, slot + 1, cl1, 1, seg1, L, M, 0u, w1, lane, nt1)) K.have[slot + 1] = 1;
                        }
                    }
                } else K.want = nt;  // the request is repeated in a later iteration, once the pool can take it
                wave_fence();
                return 2;
            }
        }
    }
    wave_fence();
    return 0;
}

// advance a deferred unit as far as possible; true when it is finished
template <bool PE>
__device__ __forceinline__ bool heavy_advance(const AlignArgs &A, const HeavyArgs &H, HState *S, uint32_t hidx, const BlockLds &BL, MateLds &LA, MateLds &LB, Mate &MA,
                              Mate &MB, const UnitSlabs &U, uint32_t &pcnt_reg, HCursor &K, int lane, Counters &C, u64 *lds_sort)
{
    const DevParams &P = A.P;
    if (PE && !MA.u->filtered && !MB.u->filtered) {
        const int maxi = max(MA.u->max_snp, MB.u->max_snp);  // PairAlign::RunAlign (pairs.cpp:163-172)
        for (;;) {
            if (K.level > maxi) return true;
            // the two SnpAlign calls of a level (pairs.cpp:165-166) touch disjoint state: both mates go as far as they can in the same
            // visit and publish their windows together — half the visits and passes of taking them in turn.  Bit m of K.sub: mate m's
            // call of this level is complete.
            bool waiting = false;
            for (int m = 0; m < 2; m++) {
                if ((K.sub >> m) & 1) continue;
                Mate &M = m ? MB : MA;
                if (K.level < M.u->seedseg && snp_align_heavy<PE>(A, H, S, hidx, BL, m ? LB : LA, M, m ? U.SB : U.SA, K.level, K, m, lane, C) == 2) { waiting = true; continue; }
                K.sub |= 1 << m; K.have[2 * m] = 0; K.have[2 * m + 1] = 0;
            }
            if (waiting) {
                if (K.have[0] | K.have[1] | K.have[2] | K.have[3]) K.want = 0;  // a published window must be replayed in the next pass (its records live one pass): never parked
                return false;
            }
            CAT_BEGIN(A);
            const int np_ = pair_level_post(P, MA, MB, U, pcnt_reg, K.level, lane, lds_sort);
            CAT_END(A, 3);
            if (np_ > 0) { K.paired = K.level + 1; return true; }
            K.level++; K.sub = 0;
            for (int m = 0; m < 2; m++) { K.orient[m] = 0; K.c[m] = 0; K.W[m] = HS_WIN0; K.have[2 * m] = 0; K.have[2 * m + 1] = 0; }
        }
    }
    // SingleAlign::RunAlign (align.cpp:445-449) for each surviving mate in turn; K.sub selects the mate (and its cursor slot)
    for (;;) {
        if (K.sub > (PE ? 1 : 0)) return true;
        const bool second = K.sub == 1;
        const int ms = second ? 1 : 0;
        Mate &M = second ? MB : MA;
        if (M.u->filtered || K.level >= M.u->seedseg) { K.sub++; K.level = 0; continue; }
        if (snp_align_heavy<PE>(A, H, S, hidx, BL, second ? LB : LA, M, second ? U.SB : U.SA, K.level, K, ms, lane, C) == 2) return false;
        const u64 nz = P.rrbs ? 0ull : bsx_ballot(M.cnt_reg != 0 && (lane & 15) <= K.level && lane < 32);  // RRBS runs all rounds (align.cpp:448)
        if (nz) { K.sub++; K.level = 0; }
        else K.level++;
        K.orient[ms] = 0; K.c[ms] = 0; K.W[ms] = HS_WIN0; K.have[2 * ms] = 0; K.have[2 * ms + 1] = 0;
    }
}

#ifndef BSX_HCTRL_WAVES_SE
#define BSX_HCTRL_WAVES_SE 1  /* the same for single-end batches */
#endif
#ifndef BSX_HCTRL_WAVES
#define BSX_HCTRL_WAVES 1  /* waves per SIMD the control kernel's register budget allows (1 = 512 registers) */
#endif
template <bool PE>
__global__ __launch_bounds__(256, PE ? BSX_HCTRL_WAVES : BSX_HCTRL_WAVES_SE) void k_hctrl(AlignArgs A_, HeavyArgs H_)
{
    __shared__ BlockLds BL;
    __shared__ WaveLds<PE> WL[4];
    __shared__ u64 SORTBUF[4][BSX_LDS_SORT];
    // The helpers called from here (scan, replay, prepare / finish, state save / restore) are real calls that take the arguments,
    // the cursor, the counters and the slab pointers by reference: as private objects they would live in scratch memory — 256 bytes
    // and four cache lines per scalar access, in a kernel that is one chain of dependent accesses.  All of them are wave-uniform:
    // one copy per block (arguments) or per wave in LDS instead.
    __shared__ AlignArgs As;
    __shared__ HeavyArgs Hs;
    __shared__ HCursor KS[4];
    __shared__ Counters CS[4];
    __shared__ UnitSlabs US[4];
    __shared__ uint32_t PEND[4][32];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < sizeof(AlignArgs) / 4; i += 256) ((uint32_t *)&As)[i] = ((const uint32_t *)&A_)[i];
    for (uint32_t i = threadIdx.x; i < sizeof(HeavyArgs) / 4; i += 256) ((uint32_t *)&Hs)[i] = ((const uint32_t *)&H_)[i];
    __syncthreads();
    const AlignArgs &A = As;
    const HeavyArgs &H = Hs;
    init_block_lds(A.P, BL, threadIdx.x, 256);
    __syncthreads();
    MateLds &LA = WL[wv].mate[0];
    MateLds &LB = WL[wv].mate[PE ? 1 : 0];
    Counters Cflush = {0, 0, 0, 0};
    u64 n_units_done = 0, n_aligned = 0, n_aligned_pairs = 0;
    const uint32_t n_active_in = H.fresh ? H.n_active_in : rfl(*H.n_active_in_ptr);  // later passes: count left by the previous pass
    // One word takes ~88 atomics per microsecond and an RRBS pass visits 10^5 units: a wave takes queue entries in chunks (one while units are few: the pass
    // then ends with its longest visit, not with a wave's leftover chunk) and hands in the units it leaves active 32 at a time (PEND).
#ifndef BSX_QCHUNK_DIV
#define BSX_QCHUNK_DIV 128u
#endif
    const uint32_t q_chunk = BSX_HCTRL_BATCH ? max(1u, min(16u, n_active_in / (gridDim.x * BSX_QCHUNK_DIV))) : 1u;   // (RRBS: 175 short visits per wave and pass, chunks of 5; C5: 25 long ones, one at a time)
    uint32_t q_next = 0, q_end = 0, n_pend = 0;
    uint32_t *const pend = PEND[wv];
#define HCTRL_PEND_FLUSH() do { if (n_pend) { uint32_t b_ = 0; if (lane == 0) b_ = atomicAdd(H.n_active_out, n_pend); b_ = rfl(b_); if ((uint32_t)lane < n_pend) H.active_out[b_ + (uint32_t)lane] = pend[lane]; n_pend = 0; wave_fence(); } } while (0)
#define HCTRL_PEND_PUSH(x) do { if (lane == 0) pend[n_pend] = (x); n_pend++; wave_fence(); if (n_pend == (BSX_HCTRL_BATCH ? 32u : 1u)) HCTRL_PEND_FLUSH(); } while (0)
    for (;;) {
        if (q_next == q_end) {
            uint32_t i0 = 0;
            if (lane == 0) i0 = atomicAdd(H.queue, q_chunk);
            q_next = rfl(i0); q_end = min(q_next + q_chunk, n_active_in);
            if (q_next >= n_active_in) break;
        }
        const uint32_t i = q_next++;
        // (later passes take the list back to front: a unit whose visit ended last in the previous pass — a long visit — was appended
        //  last; starting those first keeps the pass from waiting for one long visit that began when all the others were done)
        const uint32_t hidx = H.fresh ? H.hidx_base + i : rfl(H.active_in[n_active_in - 1u - i]);
        const uint32_t unit = rfl(A.heavy_list[H.list_base + hidx]);
        HState *S = &H.state[hidx];
        if (!H.fresh) {
            // a unit whose last request was refused stays parked (no state restore / save) while the pool cannot take it:
            // n_tasks only grows during a pass, so the reservation below would be refused again
            const uint32_t want = rfl(S->want);
            if (want && rfl(__hip_atomic_load(H.n_tasks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) + want > H.task_cap) {
                HCTRL_PEND_PUSH(hidx);
                continue;
            }
        }
        // (the slabs of deferred units carry the ordinary, small duplicate set even where the main kernel's are large — single-end
        //  RRBS —: there are too many deferred units for 4 MB each; a unit that overflows it is redone by the main kernel, below)
        uint8_t *slab = A.debug ? A.scratch + (size_t)unit * A.slab_bytes : H.slabs + (size_t)hidx * A.hslab_bytes;
        UnitSlabs &U = US[wv];
        U = A.debug ? carve_slab(slab, (uint32_t)A.P.max_snp_num + 1, A.rowcap, PE, A.kcap, A.hbits)
                    : carve_slab(slab, (uint32_t)A.P.max_snp_num + 1, A.rowcap, PE, A.hkcap, A.hhbits);
        Mate MA, MB;
        MA.u = lds_mate(&LA.u); MB.u = PE ? lds_mate(&LB.u) : lds_mate(&LA.u2);
        Counters &C = CS[wv];
        C.n_lookup = 0; C.n_cand = 0; C.sum_w = 0; C.n_orient = 0;
        HCursor &K = KS[wv];
        K.n_active = n_active_in; K.want = 0;
        for (int k_ = 0; k_ < 8; k_++) { K.vc[k_] = 0; K.vn[k_] = 0; }
        uint32_t pcnt_reg = 0;
        const u64 cat_prep0 = A.dbg_cat ? __builtin_readcyclecounter() : 0;
        if (H.fresh) {
            unit_prepare<PE, true>(A, BL, LA, LB, MA, MB, unit, lane, C);
            K.level = 0; K.sub = 0; K.paired = 0;
            for (int m_ = 0; m_ < 2; m_++) { K.orient[m_] = 0; K.have[2 * m_] = 0; K.have[2 * m_ + 1] = 0; K.c[m_] = 0; K.W[m_] = HS_WIN0; }
        } else {
            load_mate(S->mate[0], MA, LA, lane);
            if (PE) load_mate(S->mate[1], MB, LB, lane); else MB = MA;
            C = S->C;
            C.n_lookup = (u64)rfl((uint32_t)(C.n_lookup >> 32)) << 32 | rfl((uint32_t)C.n_lookup); C.n_cand = (u64)rfl((uint32_t)(C.n_cand >> 32)) << 32 | rfl((uint32_t)C.n_cand);
            C.sum_w = (u64)rfl((uint32_t)(C.sum_w >> 32)) << 32 | rfl((uint32_t)C.sum_w); C.n_orient = (u64)rfl((uint32_t)(C.n_orient >> 32)) << 32 | rfl((uint32_t)C.n_orient);
            pcnt_reg = S->pcnt_reg[lane];
            K.level = (int)rfl((uint32_t)S->level); K.sub = (int)rfl((uint32_t)S->sub); K.paired = (int)rfl((uint32_t)S->paired);
            for (int m_ = 0; m_ < 2; m_++) {
                K.orient[m_] = (int)rfl((uint32_t)S->orient[m_]); K.have[2 * m_] = (int)rfl((uint32_t)S->have[2 * m_]); K.have[2 * m_ + 1] = (int)rfl((uint32_t)S->have[2 * m_ + 1]); K.c[m_] = rfl(S->c[m_]); K.W[m_] = rfl(S->W[m_]);
            }
        }
        if (A.dbg_cat && lane == 0) { const u64 d_ = (u64)__builtin_readcyclecounter() - cat_prep0; atomicAdd((u64 *)&A.dbg_cat[0], d_); atomicMax((u64 *)&A.dbg_cat[8], d_); }
        const u64 cat_adv0 = A.dbg_cat ? __builtin_readcyclecounter() : 0;
        const bool done = heavy_advance<PE>(A, H, S, hidx, BL, LA, LB, MA, MB, U, pcnt_reg, K, lane, C, SORTBUF[threadIdx.x >> 6]);
        if (A.dbg_cat && lane == 0) {
            const u64 d_ = (u64)__builtin_readcyclecounter() - cat_adv0;
            atomicAdd((u64 *)&A.dbg_cat[6], d_);
            if (atomicMax((u64 *)&A.dbg_cat[14], d_) < d_)  // the longest visit so far: leave its break-down (racy, diagnostics only)
                for (int k_ = 0; k_ < 6; k_++) A.dbg_cat[16 + k_] = (K.vc[k_] << 16) | min(K.vn[k_], 0xffffu);
        }
        const u64 cat_fin0 = A.dbg_cat ? __builtin_readcyclecounter() : 0;
        if (done && ((MA.u->flags | (PE ? MB.u->flags : 0u)) & 4u)) {
            // the small duplicate set of this unit's heavy slab overflowed (single-end RRBS: coordinates its fragment filter rejects
            // are remembered too): its records are not written; the main kernel redoes it alone with its large set, undeferred
            forget_keys(MA, U.SA, lane); if (PE) forget_keys(MB, U.SB, lane);
            if (lane == 0) A.redo_list[atomicAdd(A.redo_count, 1u)] = unit;
        } else if (done) {
            unit_finish<PE>(A, LA, LB, MA, MB, U, pcnt_reg, K.paired, unit, lane, n_aligned, n_aligned_pairs);
            Cflush.n_lookup += C.n_lookup; Cflush.n_cand += C.n_cand; Cflush.sum_w += C.sum_w; Cflush.n_orient += C.n_orient;
            n_units_done++;
        } else {
            save_mate(S->mate[0], MA, LA, lane);
            if (PE) save_mate(S->mate[1], MB, LB, lane);
            S->pcnt_reg[lane] = pcnt_reg;
            if (lane == 0) {
                S->want = K.want; S->C = C; S->level = K.level; S->sub = K.sub; S->paired = K.paired;
                for (int m_ = 0; m_ < 2; m_++) {
                    S->orient[m_] = K.orient[m_]; S->have[2 * m_] = K.have[2 * m_]; S->have[2 * m_ + 1] = K.have[2 * m_ + 1]; S->c[m_] = K.c[m_]; S->W[m_] = K.W[m_];
                }
            }
            HCTRL_PEND_PUSH(hidx);
        }
        if (A.dbg_cat && lane == 0) { const u64 d_ = (u64)__builtin_readcyclecounter() - cat_fin0; atomicAdd((u64 *)&A.dbg_cat[4], d_); atomicMax((u64 *)&A.dbg_cat[12], d_); }
        wave_fence();
    }
    HCTRL_PEND_FLUSH();
#undef HCTRL_PEND_PUSH
#undef HCTRL_PEND_FLUSH
    if (lane == 0) flush_counters(A, Cflush, n_units_done, n_aligned, n_aligned_pairs);
}

// every wave of the chip evaluates tasks: HS_TASK consecutive candidates of one published list window
#ifndef BSX_HSCAN_WAVES
#define BSX_HSCAN_WAVES 6  /* waves per SIMD the register budget is set for */
#endif
#ifndef BSX_HSCAN_WPB
#define BSX_HSCAN_WPB 2  /* waves (= tasks) per block: 2 measured best (1: 104.5, 2: 101.3, 4: 102.9, 8: 106.2, 16: 116.6 ms per step) */
#endif

struct ScanAcc { uint32_t c1, f5, nv; };  // per wave (scalar registers, counted with s_bcnt1 on the compare masks): candidates with w0ref > thres / evaluated in full with w01ref <= thres / (RRBS) candidates at all
// popcount(x) + acc in one instruction (the compiler prefers separate counts and a v_add3)
__device__ __forceinline__ uint32_t popc_acc(uint32_t x, uint32_t acc)
{
    uint32_t d;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(acc));
    return d;
}

// ---------------------------------------------------------------------------------------------------------------
// k_hscan on bit planes (the packed-copy form of rounds 2-3 is in the history).  The candidate's reference comes from the PLANE copy (DevParams::refplane:
// per 32 nt a {low bits, high bits} pair) and is compared where it lies; the READ is what gets shifted — once per task, into a
// per-wave LDS table of its 32 possible shifts (the reference's own scheme, align.cpp:107-161, on 32-nt words).  Per 32 nt a
// candidate then costs three v_bitop3 and one v_bcnt (bsx_plane_mismatch) — no funnel shifts, no per-candidate shift amounts
// or masks — against five instructions per 16 nt on the packed copy.
//   table  PT[q][s] (q = 0..9, 8 bytes each; s = candidate position mod 32): the 20 dwords of shift s are
//          X0 Y0 | M0 M1 | X1 Y1 | B X2 | Y2 M2 | X3 Y3 | M3 X4 | Y4 M4 | X5 Y5 | M5 -      (Xj / Yj / Mj: low / high / not-N plane of frame
//          word j, B: bsx_plane_bmask).  ds_read_b64 serves 32 lanes per LDS cycle over 64 banks: with the 32 shifts of one q in 32
//          different 8-byte units no pattern of shifts conflicts (ds_read_b128 / ds_read2_b64 serve 16 lanes over 16 units:
//          shifts s and s + 16 would collide in nearly every group).  The reads are inline assembly for that reason — the compiler
//          would merge neighbouring ds_read_b64 into ds_read2_b64.
//   stage 1 (every candidate): one 16-byte gather = pairs k, k+1 (k = position >> 5) against frame words 0, 1: read nt [0, 64 - s),
//          33..64 of them.  That settles the reference's first early-out (w0ref = word 0 + word 1 & B) for every candidate.
//   stage 2 (candidates still within the threshold): pairs k+2 .. k+5 against words 2..5, through the per-wave FIFO as before
//          (8-byte items: position, partial count | ordinal | strand).
// Work accounting as before: w0ref <= p64 <= w01ref, so  words = 2 n - #(w0ref > thres) + 3 #(w01ref <= thres).
// ---------------------------------------------------------------------------------------------------------------
#define HP_QCAP 128u  /* FIFO slots per wave (8 bytes each): at most 63 left over + one chunk of 64 pushed between drains */
#define HP_PAIRS 10u
typedef unsigned long long q64;
__device__ __forceinline__ uint32_t qlo(q64 v) { return (uint32_t)v; }
__device__ __forceinline__ uint32_t qhi(q64 v) { return (uint32_t)(v >> 32); }
// table pairs 0..3 of the shift whose first pair lies at LDS byte address `addr` (a pair of the next q lies 256 bytes further)
__device__ __forceinline__ void hp_lds4(uint32_t addr, q64 &a, q64 &b, q64 &c, q64 &d)
{
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:256\n\tds_read_b64 %2, %4 offset:512\n\tds_read_b64 %3, %4 offset:768"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(addr));
}
// pairs 3..9
__device__ __forceinline__ void hp_lds7(uint32_t addr, q64 (&q)[7])
{
    asm volatile("ds_read_b64 %0, %7 offset:768\n\tds_read_b64 %1, %7 offset:1024\n\tds_read_b64 %2, %7 offset:1280\n\tds_read_b64 %3, %7 offset:1536\n\t"
                 "ds_read_b64 %4, %7 offset:1792\n\tds_read_b64 %5, %7 offset:2048\n\tds_read_b64 %6, %7 offset:2304"
                 : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]) : "v"(addr));
}
// (the compiler does not see the loads above: every value goes through the wait that makes it valid)
#define HP_WAIT4N(n, a, b, c, d) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))

// (x << 3) + c in one instruction (the compiler turns ((p >> 5) << 3) + c into shift, mask, add)
__device__ __forceinline__ uint32_t shl3_add(uint32_t x, uint32_t c) { uint32_t d; asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(d) : "v"(x), "v"(c)); return d; }
__device__ __forceinline__ uint32_t hp_boff(uint32_t p, uint32_t strand_off) { return shl3_add(p >> 5, strand_off); }           // byte offset of the pair that holds position p
__device__ __forceinline__ uint32_t hp_taddr(uint32_t tbase, uint32_t p) { return shl3_add(p & 31u, tbase); }                   // LDS address of table pair 0 of shift p mod 32

struct PlaneCtx {
    const uint8_t *plane;    // plane copy, forward strand copy first
    uint32_t rc_off;         // byte offset of the rc strand copy
    uint32_t tbase;          // LDS byte address of this wave's table
    uint2 *Q;                // this wave's FIFO
    uint32_t qh, qn;         // head slot, items queued (wave-uniform)
    uint32_t thres0, nsurv;
    int nW, lane;            // nW: frame words a read of this length can reach (2..6)
    bool overflow;
    HTaskOut *o;
    ScanAcc acc;
};

// the table of one read: lane s < 32 writes the 20 dwords of shift s
__device__ __forceinline__ void hp_build_table(const ListReq &R, uint2 *T, int lane)
{
    uint32_t x[7], y[7], m[7];  // [j + 1] = plane word j of the read; words -1 and 5 are empty
    x[0] = y[0] = m[0] = 0; x[6] = y[6] = m[6] = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) { x[j + 1] = rfl(R.px[j]); y[j + 1] = rfl(R.py[j]); m[j + 1] = rfl(R.pm[j]); }
    if (lane < 32) {
        const uint32_t s = (uint32_t)lane;
        uint32_t X[6], Y[6], M[6];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            X[j] = __builtin_amdgcn_alignbit(x[j], x[j + 1], s); Y[j] = __builtin_amdgcn_alignbit(y[j], y[j + 1], s); M[j] = __builtin_amdgcn_alignbit(m[j], m[j + 1], s);
        }
        T[0 * 32 + s] = make_uint2(X[0], Y[0]); T[1 * 32 + s] = make_uint2(M[0], M[1]); T[2 * 32 + s] = make_uint2(X[1], Y[1]);
        T[3 * 32 + s] = make_uint2(bsx_plane_bmask(s), X[2]); T[4 * 32 + s] = make_uint2(Y[2], M[2]); T[5 * 32 + s] = make_uint2(X[3], Y[3]);
        T[6 * 32 + s] = make_uint2(M[3], X[4]); T[7 * 32 + s] = make_uint2(Y[4], M[4]); T[8 * 32 + s] = make_uint2(X[5], Y[5]);
        T[9 * 32 + s] = make_uint2(M[5], 0u);
    }
}

// stage 2 for the first n (<= 64) queued candidates.  STATS: also the reference's second early-out class (work counters, AlignArgs::work_counters)
template <bool STATS>
__device__ __forceinline__ void hp_drain(PlaneCtx &X, uint32_t n)
{
    const bool act = (uint32_t)X.lane < n;
    const uint2 it = X.Q[(X.qh + (uint32_t)X.lane) & (HP_QCAP - 1)];  // x position, y p64 | ordinal << 8 | strand << 31
    X.qh = (X.qh + n) & (HP_QCAP - 1); X.qn -= n;
    const uint32_t p = it.x;
    const uint32_t boff = hp_boff(p, (uint32_t)((int32_t)it.y >> 31) & X.rc_off);
    const uint32_t taddr = hp_taddr(X.tbase, p);
    q64 q[7];
    hp_lds7(taddr, q);
    U4 g1, g2;
    g1.a = g1.b = g1.c = g1.d = 0; g2.a = g2.b = g2.c = g2.d = 0;
    if (act && X.nW > 2) g1 = *reinterpret_cast<const U4 *>(X.plane + boff + 16);
    if (act && X.nW > 4) g2 = *reinterpret_cast<const U4 *>(X.plane + boff + 32);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]));
    const uint32_t p64 = it.y & 0xffu;
    const uint32_t mm2 = bsx_plane_mismatch(g1.a, g1.b, qhi(q[0]), qlo(q[1]), qhi(q[1]));
    uint32_t tot = popc_acc(mm2, p64);
    tot = popc_acc(bsx_plane_mismatch(g1.c, g1.d, qlo(q[2]), qhi(q[2]), qlo(q[3])), tot);
    tot = popc_acc(bsx_plane_mismatch(g2.a, g2.b, qhi(q[3]), qlo(q[4]), qhi(q[4])), tot);
    tot = popc_acc(bsx_plane_mismatch(g2.c, g2.d, qlo(q[5]), qhi(q[5]), qlo(q[6])), tot);
    if (STATS) {
        const uint32_t w01ref = popc_acc(mm2 & qlo(q[0]), p64);
        X.acc.f5 += (uint32_t)__builtin_popcountll(bsx_ballot(act && w01ref <= X.thres0));
    }
    // (chromosome / end-of-sequence test and hit coordinates are left to the control kernel's replay: the record carries
    //  the strand copy and the global position)
    const bool pass = act && tot <= X.thres0;
    const u64 m = bsx_ballot(pass);
    if (m) {
        const uint32_t pos = X.nsurv + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (pass && pos < HS_SCAP) { SurvRec r; r.w_ord = tot | ((it.y & 0x7fffff00u)); r.hchr = it.y >> 31; r.hloc = p; r.hkey = 0; X.o->surv[pos] = r; }
        X.nsurv += (uint32_t)__builtin_popcountll(m);
        if (X.nsurv > HS_SCAP) X.overflow = true;
    }
}

// stage 1 for one chunk: lane l holds one candidate (p its position, r0 its first two reference pairs, q0..q3 the first four
// table pairs of its shift); candidates still within the threshold go into the FIFO.  MASKED: some lanes hold no candidate.
template <bool MASKED, bool STATS>
__device__ __forceinline__ void hp_eval(PlaneCtx &X, const U4 r0, q64 q0, q64 q1, q64 q2, q64 q3, uint32_t p, bool valid, uint32_t tag)
{
    const uint32_t mm0 = bsx_plane_mismatch(r0.a, r0.b, qlo(q0), qhi(q0), qlo(q1));
    const uint32_t mm1 = bsx_plane_mismatch(r0.c, r0.d, qlo(q2), qhi(q2), qhi(q1));
    const uint32_t c0 = __popc(mm0);
    const uint32_t p64 = popc_acc(mm1, c0);
    const bool need = (!MASKED || valid) && p64 <= X.thres0;
    if (STATS) {
        const uint32_t w0ref = popc_acc(mm1 & qlo(q3), c0);
        X.acc.c1 += (uint32_t)__builtin_popcountll(bsx_ballot((!MASKED || valid) && w0ref > X.thres0));
    }
    const u64 nm = bsx_ballot(need);
    if (nm) {
        if (need) {
            const uint32_t pos = X.qh + X.qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(nm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nm, 0u));
            X.Q[pos & (HP_QCAP - 1)] = make_uint2(p, p64 | tag);
        }
        X.qn += (uint32_t)__builtin_popcountll(nm);
    }
    while (X.qn >= 64 && !X.overflow) hp_drain<STATS>(X, 64);  // (FIFO writes and reads of a wave are ordered: same wave, same LDS)
}

// stage 1 for 256 consecutive candidates of one WGBS sub-range (see hscan_step: entries one step ahead)
template <bool FULL, bool STATS>
__device__ __forceinline__ void hp_step(PlaneCtx &X, uint32_t (&e)[4], const uint32_t *__restrict__ nextq, uint32_t ref_off, uint32_t h, uint32_t n_here, uint32_t ord0,
                                        uint32_t strand)
{
    const int lane = X.lane;
    uint32_t p[4], boff[4], tag[4];
    bool valid[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        valid[u] = FULL || (uint32_t)(u * 64 + lane) < n_here;
        p[u] = (FULL || valid[u]) ? e[u] + h : 16u;
        boff[u] = hp_boff(p[u], ref_off);
        tag[u] = (ord0 + (uint32_t)(u * 64 + lane)) << 8 | strand << 31;
    }
    U4 r0[4];
    q64 q[2][4];
#pragma unroll
    for (int u = 0; u < 4; u++) r0[u] = *reinterpret_cast<const U4 *>(X.plane + boff[u]);
    if (FULL) {
#pragma unroll
        for (int u = 0; u < 4; u++) e[u] = nextq[u * 64];
    }
    // table reads one chunk ahead (two register sets): LDS operations of a wave complete in order, so once all but the four youngest
    // are done the older chunk's pairs are there, whatever the compiler issued in between
    hp_lds4(hp_taddr(X.tbase, p[0]), q[0][0], q[0][1], q[0][2], q[0][3]);
#pragma unroll
    for (int u = 0; u < 4; u++) {
        if (u < 3) { hp_lds4(hp_taddr(X.tbase, p[u + 1]), q[(u + 1) & 1][0], q[(u + 1) & 1][1], q[(u + 1) & 1][2], q[(u + 1) & 1][3]); HP_WAIT4N(4, q[u & 1][0], q[u & 1][1], q[u & 1][2], q[u & 1][3]); }
        else HP_WAIT4N(0, q[u & 1][0], q[u & 1][1], q[u & 1][2], q[u & 1][3]);
        hp_eval<!FULL, STATS>(X, r0[u], q[u & 1][0], q[u & 1][1], q[u & 1][2], q[u & 1][3], p[u], valid[u], tag[u]);
    }
}

// one scan task on one wave: candidates [c0, c0 + n) of the task's list, survivors in list order and the work counters into its output record
// (TABw / PTw / Qw: this wave's sub-range table, read table and FIFO in LDS)
template <bool STATS>
__device__ __forceinline__ void hp_task(const AlignArgs &A, const HeavyArgs &H, uint32_t t, int lane, uint32_t (&TABw)[4][32], uint2 *PTw, uint2 *Qw)
{
    const DevParams &P = A.P;
    const uint32_t hraw = rfl(H.tasks[t].h), hidx = hraw & 0x3fffffffu, tc0 = rfl(H.tasks[t].c0), tn = rfl(H.tasks[t].n);
    HTaskOut *o = &H.tout[t];
    if (tn == 0) {  // slot neutralised by a refused request: its unit has not published a list (ListReq may be stale)
        if (lane == 0) { o->count = 0; o->overflow = 0; o->acc[0] = o->acc[1] = o->acc[2] = o->acc[3] = 0; o->c0 = 0; o->n = 0; }
        return;
    }
    const ListReq &R = H.state[hidx].req[hraw >> 30];
    if (lane < 32) { TABw[0][lane] = R.sub_pre[lane]; TABw[1][lane] = R.sub_n[lane]; TABw[2][lane] = R.sub_base[lane]; TABw[3][lane] = R.sub_h[lane]; }
    hp_build_table(R, PTw, lane);
    const uint32_t nsub = min(rfl(R.nsub), 32u);
    wave_fence();
    PlaneCtx X;
    X.plane = reinterpret_cast<const uint8_t *>(P.refplane); X.rc_off = P.plane_rc_off;
    X.tbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint2 *)PTw;
    X.Q = Qw; X.qh = 0; X.qn = 0; X.thres0 = rfl(R.thres); X.nsurv = 0; X.nW = (int)((rfl(R.len) + 31u + 31u) >> 5); X.lane = lane;
    X.overflow = false; X.o = o; X.acc.c1 = 0; X.acc.f5 = 0; X.acc.nv = 0;
    const uint32_t c_end = tc0 + tn;
    // (WGBS lists only: RRBS lists — one bucket of {tag, loc} pairs — always go to k_hscan_shared, bsx_api.hip)
    for (uint32_t sidx = 0; sidx < nsub && !X.overflow; sidx++) {
        const uint32_t ps = rfl(TABw[0][sidx]), ns = rfl(TABw[1][sidx]);
        const uint32_t lo = max(tc0, ps), hi = min(c_end, ps + ns);
        if (lo >= hi) continue;
        const uint32_t *ent = P.entries + rfl(TABw[2][sidx]);
        const uint32_t h = rfl(TABw[3][sidx]), strand = sidx & 1;
        const uint32_t ref_off = strand ? X.rc_off : 0u;
        uint32_t cb = lo;
        uint32_t e[4];
#pragma unroll
        for (int u = 0; u < 4; u++) e[u] = (ent + (cb - ps) + lane)[u * 64];
        for (; cb + 256 <= hi && !X.overflow; cb += 256) hp_step<true, STATS>(X, e, ent + (cb + 256 - ps) + lane, ref_off, h, 256, cb - tc0, strand);
        if (cb < hi && !X.overflow) hp_step<false, STATS>(X, e, nullptr, ref_off, h, hi - cb, cb - tc0, strand);
    }
    while (X.qn && !X.overflow) hp_drain<STATS>(X, min(X.qn, 64u));
    const uint32_t n1 = X.acc.c1, n5 = X.acc.f5;
    const uint32_t n_cand = tn;
    const uint32_t words = STATS ? 2u * n_cand - n1 + 3u * n5 : 0u;  // 1, 2 or 5 words per candidate (see above); without the work counters: not known
    if (lane == 0) {
        o->count = min(X.nsurv, (uint32_t)HS_SCAP); o->overflow = X.overflow ? 1 : 0; o->acc[0] = n_cand; o->acc[1] = words; /* (overflowed: the first HS_SCAP survivors, in list order) */ o->acc[2] = 0; o->acc[3] = 0; o->c0 = tc0; o->n = tn;
        if (!X.overflow) {  // work of the scan kernel (incl. speculation); an overflowed task is redone by the control kernel.
            // Millions of tasks per batch: one counter word takes ~88 atomics per microsecond, so these statistics are
            // sharded over 64 cache lines (summed by bsx_batch_counters) instead of being added to four hot words
            u64 *sh = (u64 *)A.scan_stats + (size_t)(blockIdx.x & 63u) * 8;
            atomicAdd((u64 *)&sh[0], (u64)n_cand);
            if (STATS) { atomicAdd((u64 *)&sh[1], (u64)words); atomicAdd((u64 *)&sh[2], (u64)n1); atomicAdd((u64 *)&sh[3], (u64)n5); }
        }
    }
}

template <bool STATS>
__global__ __launch_bounds__(64 * BSX_HSCAN_WPB, BSX_HSCAN_WAVES) void k_hscan(AlignArgs A, HeavyArgs H)
{
    __shared__ uint32_t TAB[BSX_HSCAN_WPB][4][32];
    __shared__ uint2 PT[BSX_HSCAN_WPB][HP_PAIRS * 32];   // the read of each wave's task at its 32 shifts
    __shared__ uint2 QBUF[BSX_HSCAN_WPB][HP_QCAP];
    const int lane = threadIdx.x & 63, wv = (int)rfl(threadIdx.x >> 6);  // (the wave number as a scalar: what depends on it stays wave-uniform for the compiler)
    // one task per wave and sweep, no queue: the blocks of a pass retire one by one, so the control kernel of the other unit
    // group (high-priority stream) finds free slots while this kernel is still running
    const uint32_t n_tasks = min(*H.n_tasks, H.task_cap);
    // Tasks are taken in key order (= by the index entries they walk): the reads that walk the same giant bucket then do
    // so at the same time, and each line of entries / reference is fetched from memory once for all of them.
    // The host does not know the count: it sizes the grid for the whole task pool (blocks beyond the tasks of the pass leave at once)
    // or, in the tail of a batch, for a few thousand tasks — then a block sweeps over the order with the stride of the grid.
    const uint32_t nvb = (n_tasks + BSX_HSCAN_WPB - 1) / BSX_HSCAN_WPB;
    for (uint32_t vb = blockIdx.x;; vb += gridDim.x) {
        uint32_t b_;
        const int st_ = bsx_order_block(vb, nvb, H.xcd_map, b_);
        if (st_ == 2) break;
        if (st_ == 1) continue;
        const uint32_t slot = b_ * BSX_HSCAN_WPB + (uint32_t)wv;
        if (slot < n_tasks) hp_task<STATS>(A, H, H.order ? rfl(H.order[slot]) : slot, lane, TAB[wv], PT[wv], QBUF[wv]);
        wave_fence();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_hscan_same — WGBS: the reads that walk one window of a giant bucket WITH THE SAME READ OFFSET share the fetch and the shift of
// the candidates' reference.  publish_window cuts large sub-ranges on the grid of absolute index entries, so the tasks of all the
// reads that walk a bucket cover identical entry ranges; a read reaches the bucket's seed at one of (segments x index interval)
// offsets h, and the tasks of equal (first entry, length, strand copy, h, read words) are what a group is made of: for those the
// candidate positions entry + h are the same numbers.  The task order (bsx_launch_task_order, spread) puts the tasks of one window
// and offset class next to each other; a block takes 64 consecutive tasks, every wave splits them into the same groups of up to
// HG_R tasks (any lanes, not only neighbours) and takes every HG_WAVES-th group.  Per group and step of HG_C chunks of 64 candidates
//   * once: the entries, three 16-byte gathers per candidate (pairs k .. k + 5 of the plane copy, k = (position - 1) >> 5) and the
//     funnel shift into the READ's frame (two v_alignbit per 32 nt);
//   * per read: its plane words from LDS — ONE broadcast read of the row per step, not per chunk: a 16-byte LDS read of a wave
//     returns 1 KB whatever the addresses, 8 cycles of the CU's LDS path, and four of them per chunk (k_hscan_shared's scheme)
//     outweigh the 27 vector instructions of an evaluation — then 3 v_bitop3 + 1 v_bcnt per word and candidate, the two
//     early-out classes of the reference (align.h:189-197) and the survivors, written in list order.
// No second stage and no FIFO: on the hg38-sized workload 36-55 % of the candidates of a giant bucket are still within the
// threshold after 64 nt, so a chunk of 64 practically always holds one that needs all the words.
// Groups of one task, and tasks that span sub-ranges, go one per wave through hp_task.  Results per task are exactly k_hscan's.
// ---------------------------------------------------------------------------------------------------------------
#ifndef HG_WPB
#define HG_WPB 1        /* waves (= groups) per block: 1 measured best (1: 96.0-98.9, 2: 101.7-102.2, 4: 110.9 ms per step) — a block gives its slot back when its LAST wave ends */
#endif
#ifndef HG_R
#define HG_R 16u        /* tasks per group at most */
#endif
#ifndef HG_C
#define HG_C 4          /* chunks per step.  Round 5 (without the work counters, 2^22 pairs per batch; scan ms per step / C3 M reads/s, one box): 2 chunks with the gathers a step
                           ahead at five waves per SIMD 154.4-157.9 / 27.1-27.4; 3 without that prefetch 144.9 / 28.1; 4 without, five waves (64 B of scratch) 146.9 / 29.1; 4 without,
                           four waves 141.7-143.0 / 28.0-28.9; 5, 6 at four waves 141.6, 143.7; 8 at three 141.8 / 27.6: what a step costs per READ (its LDS row, two v_readlane,
                           the scalar bookkeeping) is spread over twice the candidates */
#endif
#ifndef HG_BIG
#define HG_BIG 1u       /* groups of this many tasks and more are started first, the others fill the tail of the launch.  1 = plain scan order, measured best:
                           from 4 tasks 103.5-104.2 against 99.4-100.3 ms per step, from 8 103.0 — the long groups of all windows at once lose the shared cache lines */
#endif
#ifndef HG_PREFETCH
#define HG_PREFETCH 0   /* 1: the gathers of a step are issued a step earlier (13 registers per chunk: only fits with two chunks per step) */
#endif
#ifndef BSX_HSAME_WAVES
#define BSX_HSAME_WAVES 4
#endif
#ifndef HG_ROWPF
#define HG_ROWPF 0      /* the next read's LDS row requested before this read is evaluated (16 registers) */
#endif
struct SameLds {
    // per read of a group (row of 20 dwords): X0 Y0 M0 X1 | Y1 M1 X2 Y2 | M2 threshold X3 Y3 | M3 X4 Y4 M4 | task, first list ordinal, -, -
    union {
        __attribute__((aligned(16))) uint32_t UW[HG_R][20];
        struct { uint32_t TAB[4][32]; uint2 PT[HP_PAIRS * 32]; uint2 Q[HP_QCAP]; } one;   // one-task path (hp_task); 4 KB per wave: k_hctrl's blocks (58 KB paired) have to fit beside 20 waves of this kernel
    } W[HG_WPB];
};
struct SameChunk { U4 r0, r1, r2; uint32_t pm1, strand; bool valid; };   // pairs (pm1 >> 5) .. + 5 of the plane copy; RRBS: the candidate's strand copy, and whether the entry passed the filters
// what a group's window is made of.  WGBS: 4-byte entries, candidate position = entry + h.  RRBS: one bucket of {tag | chromosome, position}
// pairs — the entries with ((tag ^ tag_xor) >> 16) == tag_want and position >= h are the candidates (align.cpp:187,229,263), position =
// anchor[chromosome] + (position - h), each on its own strand copy
struct SameWin { const uint32_t *ent; uint32_t n, h, tag_xor, tag_want, ref_off, rc_off; const uint32_t *anchor; const uint8_t *plane; int nwr; };
struct SameEntry { uint32_t a, b; bool in; };

template <bool RRBS>
__device__ __forceinline__ SameEntry same_entry(const SameWin &W, uint32_t idx)
{
    SameEntry e; e.a = 1024u; e.b = 0; e.in = idx < W.n;   // (a lane without a candidate takes entry 1024: any position inside the copy)
    if (RRBS) { e.a = 0; if (e.in) { const U2 v = reinterpret_cast<const U2 *>(W.ent)[idx]; e.a = v.a; e.b = v.b; } }
    else if (e.in) e.a = W.ent[idx];
    return e;
}
#ifdef BSX_SECTOR_STATS
// Diagnostic build (tools/build_variant.sh sectors -DBSX_SECTOR_STATS; BSX_SECTOR_STATS=1 at run time): which 64-byte sectors of the plane copy the group
// scan touches in a pass.  Every gather marks its sectors in a bitmap (one bit per sector: 3 MB for the hg38-sized copy) and counts its lane-sector
// touches; after every scan launch bsx_sector_pass counts and clears the bitmap: the sum over passes is the COMPULSORY traffic of the scan — what an
// ideal cache in front of one pass would still fetch — to set beside TCC_EA0_RDREQ (what the fabric served) and the touches (what L2 was asked for).
__device__ uint32_t *g_sector_bits = nullptr;
__device__ unsigned long long g_sector_touch = 0, g_sector_distinct = 0, g_sector_passes = 0;
__device__ __forceinline__ void sector_mark(uint32_t byte_off, int n16)
{
    uint32_t *bits = g_sector_bits;
    if (!bits) return;
    uint32_t prev = 0xffffffffu, touches = 0;
    for (int i = 0; i < n16; i++)
        for (uint32_t b = byte_off + 16u * (uint32_t)i; b <= byte_off + 16u * (uint32_t)i + 15u; b += 15u) {
            const uint32_t sct = b >> 6;
            if (sct == prev) continue;
            prev = sct; touches++;
            if (!((bits[sct >> 5] >> (sct & 31u)) & 1u)) atomicOr(&bits[sct >> 5], 1u << (sct & 31u));
        }
    const unsigned long long tw = (unsigned long long)wave_sum(touches);
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_sector_touch, tw);
}
__global__ __launch_bounds__(256) void k_sector_pop(uint32_t n_words)
{
    unsigned long long c = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_words; i += gridDim.x * 256) { const uint32_t w = g_sector_bits[i]; if (w) { c += (unsigned long long)__popc(w); g_sector_bits[i] = 0; } }
    for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&g_sector_distinct, c);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_sector_passes, 1ull);
}
#endif

template <bool RRBS>
__device__ __forceinline__ SameChunk same_load(const SameWin &W, const SameEntry &e)
{
    SameChunk c;
    uint32_t off = W.ref_off;
    if (RRBS) {
        const uint32_t rchr = e.a & 0xffffu;
        c.valid = e.in && ((e.a ^ W.tag_xor) >> 16) == W.tag_want && e.b >= W.h;  // mode or strand not match / underflow the start of refseq
        c.pm1 = c.valid ? W.anchor[rchr >> 1] + (e.b - W.h) - 1u : 31u;
        c.strand = rchr & 1u;
        off = c.strand ? W.rc_off : 0u;
    } else { c.valid = e.in; c.pm1 = e.a + W.h - 1u; c.strand = 0; }
    const uint8_t *src = W.plane + hp_boff(c.pm1, off);
    c.r0 = *reinterpret_cast<const U4 *>(src);
    c.r1.a = c.r1.b = c.r1.c = c.r1.d = 0; c.r2.a = c.r2.b = c.r2.c = c.r2.d = 0;
    if (W.nwr > 1) c.r1 = *reinterpret_cast<const U4 *>(src + 16);
    if (W.nwr > 3) c.r2 = *reinterpret_cast<const U4 *>(src + 32);
#ifdef BSX_SECTOR_STATS
    if (c.valid) sector_mark(hp_boff(c.pm1, off), W.nwr > 3 ? 3 : W.nwr > 1 ? 2 : 1); else sector_mark(0u, 0);
#endif
    return c;
}

// the three counts of one candidate and read (align.h:189-197): first early-out word, second, whole read.
// NWR: the read's 32-nt words if known at compile time (0: nwr_rt); PLAIN: the read has no N — only its last word is masked, the words
// before it take two v_bitop3 instead of three; SKIP: the words behind the first 64 nt only where a lane of `alive` is still within the
// threshold there (RRBS, -v 2: 99 % of a repeat family's candidates fail early; both early-out classes and the survivors are settled by then)
template <int NWR, bool PLAIN, bool SKIP, bool STATS>
__device__ __forceinline__ void same_counts(const uint32_t (&flo)[5], const uint32_t (&fhi)[5], uint32_t him, int nwr_rt, const uint4 &a0, const uint4 &a1, const uint4 &a2,
                                            const uint4 &a3, uint32_t thr, u64 alive, uint32_t &w0ref, uint32_t &w01ref, uint32_t &tot)
{
    static_assert(NWR != 0 || !PLAIN, "the plain form needs the word count");
    const int nwr = NWR ? NWR : nwr_rt;
#define SAME_MM(j, X, Y, M) ((PLAIN && (j) < NWR - 1) ? bsx_plane_mismatch_full(flo[j], fhi[j], X, Y) : bsx_plane_mismatch(flo[j], fhi[j], X, Y, M))
    const uint32_t m0 = SAME_MM(0, a0.x, a0.y, a0.z);
    const uint32_t c0 = __popc(m0);
    w0ref = STATS ? __popc(m0 & him) : 0u; tot = c0; w01ref = c0;
    if (nwr > 1) {
        const uint32_t m1 = SAME_MM(1, a0.w, a1.x, a1.y);
        tot = popc_acc(m1, c0); if (STATS) w01ref = popc_acc(m1 & him, c0);
        if (nwr > 2 && (!SKIP || (bsx_ballot(tot <= thr) & alive))) {
            tot = popc_acc(SAME_MM(2, a1.z, a1.w, a2.x), tot);
            if (nwr > 3) {
                tot = popc_acc(SAME_MM(3, a2.z, a2.w, a3.x), tot);
                if (nwr > 4) tot = popc_acc(SAME_MM(4, a3.y, a3.z, a3.w), tot);
            }
        }
    }
#undef SAME_MM
}

// one read of a group against the step's chunks: counts and survivors.  PLAIN as in same_counts; FULL: every lane of every chunk holds a candidate
// (all steps of a window but its last) — the masks need no AND with the valid lanes.  Both are decided once per read and step, outside the chunk loop
// (scan 56.2-56.6 against 57.7-57.8 ms per step with the two tests inside it)
template <int NWR, bool RRBS, bool PLAIN, bool FULL, bool STATS>
__device__ __forceinline__ void hs_eval_read(const uint32_t (&flo)[HG_C][5], const uint32_t (&fhi)[HG_C][5], const uint32_t (&him)[HG_C], const u64 (&vm)[HG_C],
                                             const uint32_t (&ordsh)[HG_C], const uint32_t (&hchr)[HG_C], const uint32_t (&hloc)[HG_C], int nwr, const uint4 &a0, const uint4 &a1,
                                             const uint4 &a2, const uint4 &a3, uint32_t thr, SurvRec *sv, uint32_t &nsk, uint32_t &add15)
{
#pragma unroll
    for (int u = 0; u < HG_C; u++) {
        uint32_t w0ref, w01ref, tot;
        same_counts<NWR, PLAIN, RRBS, STATS>(flo[u], fhi[u], him[u], nwr, a0, a1, a2, a3, thr, vm[u], w0ref, w01ref, tot);
        u64 bp = bsx_ballot(tot <= thr);
        if (!FULL) bp &= vm[u];
        if (STATS) {   // the two early-out classes of the reference (align.h:189-197): work counters only, no effect on any hit
            u64 b1 = bsx_ballot(w0ref > thr), b5 = bsx_ballot(w01ref <= thr);
            if (!FULL) { b1 &= vm[u]; b5 &= vm[u]; }
            add15 += (uint32_t)__builtin_popcountll(b1) + ((uint32_t)__builtin_popcountll(b5) << 16);
        }
        if (bp) {
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(bp >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bp, nsk));
            if (__builtin_amdgcn_inverse_ballot_w64(bp) && pos < HS_SCAP) { SurvRec r; r.w_ord = tot | ordsh[u]; r.hchr = hchr[u]; r.hloc = hloc[u]; r.hkey = 0; sv[pos] = r; }
            nsk += (uint32_t)__builtin_popcountll(bp);
        }
    }
}

// one group of K (1 .. HG_R) tasks: lane j < K holds task j's id, unit | slot and first list ordinal; they cover the n index entries
// from `key` on with read offset `hh` (RRBS: and tag filter tag_xor / tag_want).  NWR: the reads' 32-nt words where the length class has
// its own code (5: 129-160 nt, the headline configuration; 4: 97-128 nt; 3: 65-96 nt, RRBS), 0 for any length.
template <int NWR, bool RRBS, bool STATS>
__device__ __forceinline__ void hs_group(const AlignArgs &A, const HeavyArgs &H, SameLds &L, int lane, int wv, uint32_t K, uint32_t tid, uint32_t th, uint32_t tc0,
                                         uint32_t key, uint32_t n, uint32_t hh, uint32_t flags, uint32_t tag_xor, uint32_t tag_want)
{
    const DevParams &P = A.P;
    uint32_t *uw = &L.W[wv].UW[0][0];
    const int nwr = NWR ? NWR : (int)((flags >> 8) & 15u);
#ifdef BSX_SIGHIST_DUPS
    u64 rowsig = 0;
#endif
    if ((uint32_t)lane < K) {   // the read of this lane's task -> its row
        const ListReq &R = H.state[th & 0x3fffffffu].req[th >> 30];
        uint32_t *row = uw + (uint32_t)lane * 20u;
        uint32_t inner = 0xFFFFFFFFu;   // the not-N planes of the words before the last
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const int f = 3 * j, o = j < 3 ? 0 : 1;
            const uint32_t pm = R.pm[j];
            row[f + o] = R.px[j]; row[f + 1 + o] = R.py[j]; row[f + 2 + o] = pm;
            if (j < nwr - 1) inner &= pm;
        }
        row[9] = R.thres | (inner == 0xFFFFFFFFu ? 0x10000u : 0u); row[16] = tid; row[17] = tc0;
#ifdef BSX_SIGHIST_DUPS
        u64 hsh = R.thres;
#pragma unroll
        for (int j = 0; j < 5; j++) { hsh = (hsh ^ R.px[j]) * 0x9E3779B97F4A7C15ull; hsh = (hsh ^ R.py[j]) * 0xBF58476D1CE4E5B9ull; hsh = (hsh ^ R.pm[j]) * 0x94D049BB133111EBull; hsh ^= hsh >> 29; }
        rowsig = hsh;
#endif
    }
    // diagnostic build (-DBSX_SIGHIST_DUPS, with BSX_SIGHIST=1; round 5 measured 0.00 % for C3, 1.4 % for C2): members of the group whose read words
    // and threshold equal an earlier member's (they would yield the same survivors).  Not in the shipped kernel: it is bound by its vector issue.
    uint32_t n_dup = 0;
#ifdef BSX_SIGHIST_DUPS
    for (uint32_t j = 1; j < K; j++) {
        const u64 sj = ((u64)rl_u((uint32_t)(rowsig >> 32), j) << 32) | rl_u((uint32_t)rowsig, j);
        if (bsx_ballot((uint32_t)lane < j && rowsig == sj)) n_dup++;
    }
#endif
    wave_fence();
    const uint32_t strand = flags & 1u;
    SameWin W;
    W.ent = P.entries + (RRBS ? 2u * (size_t)key : (size_t)key); W.n = n; W.h = hh; W.tag_xor = tag_xor; W.tag_want = tag_want; W.rc_off = P.plane_rc_off;
    W.ref_off = strand ? P.plane_rc_off : 0u; W.anchor = P.anchor; W.plane = reinterpret_cast<const uint8_t *>(P.refplane); W.nwr = nwr;
    uint32_t c15 = 0, nsv = 0;   // lane k: counters of read k — candidates beyond the first word | five-word candidates << 16; survivors
    uint32_t n_cand = 0;         // RRBS: the entries that passed the filters
    // one step = HG_C chunks of 64 candidates.  nx: the gathers of the coming step (issued a step earlier); en: the entries of the step after it
    constexpr uint32_t STEP = 64u * HG_C;
    SameChunk nx[HG_C];
    SameEntry en[HG_C];
#if HG_PREFETCH
#pragma unroll
    for (int u = 0; u < HG_C; u++) nx[u] = same_load<RRBS>(W, same_entry<RRBS>(W, (uint32_t)(u * 64 + lane)));
#pragma unroll
    for (int u = 0; u < HG_C; u++) en[u] = same_entry<RRBS>(W, STEP + (uint32_t)(u * 64 + lane));
    // (the first step's gathers are waited for here: pending into the loop they make the compiler wait for every load in flight — the coming step's too — before each step's
    //  evaluation, and the prefetch is none: round 5's measurements of this switch were of that form)
#pragma unroll
    for (int u = 0; u < HG_C; u++) asm volatile("" :: "v"(nx[u].r0.a), "v"(nx[u].r1.a), "v"(nx[u].r2.a));
#else
#pragma unroll
    for (int u = 0; u < HG_C; u++) en[u] = same_entry<RRBS>(W, (uint32_t)(u * 64 + lane));
#endif
    for (uint32_t cb = 0; cb < n; cb += STEP) {
#if !HG_PREFETCH
#pragma unroll
        for (int u = 0; u < HG_C; u++) nx[u] = same_load<RRBS>(W, en[u]);
#pragma unroll
        for (int u = 0; u < HG_C; u++) en[u] = same_entry<RRBS>(W, cb + STEP + (uint32_t)(u * 64 + lane));
#endif
        // the candidates' reference planes in the read frame — the same for every read of the group
        uint32_t flo[HG_C][5], fhi[HG_C][5], him[HG_C], hloc[HG_C], ordsh[HG_C], hchr[HG_C];
        u64 vm[HG_C];
#pragma unroll
        for (int u = 0; u < HG_C; u++) {
            const SameChunk &c = nx[u];
            const uint32_t wd[12] = {c.r0.a, c.r0.b, c.r0.c, c.r0.d, c.r1.a, c.r1.b, c.r1.c, c.r1.d, c.r2.a, c.r2.b, c.r2.c, c.r2.d};
            const uint32_t shf = 31u - (c.pm1 & 31u);                     // 32 - ((pm1 & 31) + 1)
            him[u] = 0xFFFFFFFFu << ((c.pm1 + 1u) & 15u);                 // read nt [0, 32 - k), k = position mod 16
#pragma unroll
            for (int t = 0; t < 5; t++) { flo[u][t] = __builtin_amdgcn_alignbit(wd[2 * t], wd[2 * t + 2], shf); fhi[u][t] = __builtin_amdgcn_alignbit(wd[2 * t + 1], wd[2 * t + 3], shf); }
            hloc[u] = c.pm1 + 1u;
            hchr[u] = RRBS ? c.strand : strand;
            ordsh[u] = (cb + (uint32_t)(u * 64 + lane)) << 8;
            vm[u] = bsx_ballot(c.valid);
            if (RRBS) n_cand += (uint32_t)__builtin_popcountll(vm[u]);
        }
#if HG_PREFETCH
        if (cb + STEP < n) {   // the next step's gathers fly while this step is evaluated, and the entries of the step after it
#pragma unroll
            for (int u = 0; u < HG_C; u++) nx[u] = same_load<RRBS>(W, en[u]);
#pragma unroll
            for (int u = 0; u < HG_C; u++) en[u] = same_entry<RRBS>(W, cb + 2u * STEP + (uint32_t)(u * 64 + lane));
        }
#endif
        const bool full = !RRBS && cb + STEP <= n;   // (RRBS: the filters decide per entry)
#if HG_ROWPF
        // (the next read's row is requested before this read is evaluated: its LDS round trip — four 16-byte reads of a wave — hides behind ~50 vector instructions)
        uint4 p0, p1, p2, p3 = make_uint4(0u, 0u, 0u, 0u);
        { const uint4 *row = reinterpret_cast<const uint4 *>(uw); p0 = row[0]; p1 = row[1]; p2 = row[2]; if (nwr > 3) p3 = row[3]; }
#endif
        for (uint32_t k = 0; k < K; k++) {
            // (a 16-byte LDS read of a wave moves 1 KB, 8 cycles of the CU's LDS path: one read of the row per step, not per chunk)
#if HG_ROWPF
            const uint4 a0 = p0, a1 = p1, a2 = p2, a3 = p3;
            if (k + 1 < K) { const uint4 *row = reinterpret_cast<const uint4 *>(uw + (k + 1u) * 20u); p0 = row[0]; p1 = row[1]; p2 = row[2]; if (nwr > 3) p3 = row[3]; }
#else
            const uint4 *row = reinterpret_cast<const uint4 *>(uw + k * 20u);
            const uint4 a0 = row[0], a1 = row[1], a2 = row[2];  // X0 Y0 M0 X1 | Y1 M1 X2 Y2 | M2 threshold X3 Y3 | M3 X4 Y4 M4
            uint4 a3 = make_uint4(0u, 0u, 0u, 0u);
            if (nwr > 3) a3 = row[3];
#endif
            const uint32_t tp = rfl(a2.y), thr = tp & 0xffffu;
            const bool plain = NWR != 0 && (tp >> 16) != 0;
            SurvRec *const sv = H.tout[rl_u(tid, k)].surv;   // (wave-uniform: the address arithmetic stays on the scalar unit)
            uint32_t nsk = rl_u(nsv, k), add15 = 0;
            if (full) {
                if (plain) hs_eval_read<NWR, RRBS, NWR != 0, !RRBS, STATS>(flo, fhi, him, vm, ordsh, hchr, hloc, nwr, a0, a1, a2, a3, thr, sv, nsk, add15);
                else hs_eval_read<NWR, RRBS, false, !RRBS, STATS>(flo, fhi, him, vm, ordsh, hchr, hloc, nwr, a0, a1, a2, a3, thr, sv, nsk, add15);
            } else {
                if (plain) hs_eval_read<NWR, RRBS, NWR != 0, false, STATS>(flo, fhi, him, vm, ordsh, hchr, hloc, nwr, a0, a1, a2, a3, thr, sv, nsk, add15);
                else hs_eval_read<NWR, RRBS, false, false, STATS>(flo, fhi, him, vm, ordsh, hchr, hloc, nwr, a0, a1, a2, a3, thr, sv, nsk, add15);
            }
            if ((uint32_t)lane == k) { if (STATS) c15 += add15; nsv = nsk; }
        }
    }
    wave_fence();
    // results: lane j < K holds read j's counters
    if (!RRBS) n_cand = n;
    const bool mine = (uint32_t)lane < K;
    const uint32_t n1 = c15 & 0xffffu, n5 = c15 >> 16, ns = nsv;
    const bool ov = ns > HS_SCAP;
    if (mine) {
        HTaskOut *o = &H.tout[tid];
        o->count = min(ns, (uint32_t)HS_SCAP); o->overflow = ov ? 1 : 0; o->acc[0] = n_cand; o->acc[1] = STATS ? 2u * n_cand - n1 + 3u * n5 : 0u; o->acc[2] = 0; o->acc[3] = 0; o->c0 = tc0; o->n = n;
    }
    const bool cnt = mine && !ov;
    const uint32_t kk = (uint32_t)__builtin_popcountll(bsx_ballot(cnt));
    const uint32_t s1 = wave_sum(cnt ? n1 : 0), s5 = wave_sum(cnt ? n5 : 0);
    if (lane == 0 && kk) {   // (sharded statistics: see hp_task)
        u64 *sh = (u64 *)A.scan_stats + (size_t)((blockIdx.x * (uint32_t)HG_WPB + (uint32_t)wv) & 63u) * 8;
        atomicAdd((u64 *)&sh[0], (u64)kk * n_cand);
        if (STATS) { atomicAdd((u64 *)&sh[1], 2ull * kk * n_cand - s1 + 3ull * s5); atomicAdd((u64 *)&sh[2], (u64)s1); atomicAdd((u64 *)&sh[3], (u64)s5); }
        if (K > 1) atomicAdd((u64 *)&sh[4], (u64)kk * n_cand);   // counter 15: candidates evaluated in groups of two reads and more
        atomicAdd((u64 *)&sh[5], (u64)kk * n_cand * K); if (K >= 4) atomicAdd((u64 *)&sh[6], (u64)kk * n_cand); if (n_dup) atomicAdd((u64 *)&sh[7], (u64)n_dup * n_cand);  // diagnostics (BSX_SIGHIST)
    }
    wave_fence();
}

// Groups are formed before the scan (k_task_groups, below): behind it H.order holds the members of a group next to each other,
// H.ghead[slot] the size of the group that starts at scan slot `slot` and H.glist the slots at which groups start (their number in
// H.glist[task_cap]).  One group per wave, as k_hscan takes one task per wave: the waves of a pass retire one by one, every
// launched wave has work, and consecutive waves walk the groups of one window at the same time.
template <bool STATS>
__global__ __launch_bounds__(64 * HG_WPB, BSX_HSAME_WAVES) void k_hscan_same(AlignArgs A, HeavyArgs H)
{
    __shared__ SameLds L;
    const int lane = threadIdx.x & 63, wv = (int)rfl(threadIdx.x >> 6);
    const uint32_t n_big = min(H.glist[H.task_cap], H.task_cap), n_groups = min(n_big + H.glist[H.task_cap + 1], H.task_cap);
    const uint32_t nvb = (n_groups + HG_WPB - 1) / HG_WPB;
    for (uint32_t vb = blockIdx.x;; vb += gridDim.x) {
        uint32_t b_;
        // (pieces of 256 groups per XCD turn — about a thousand tasks, twice k_hscan's 512: 64 groups 97.7-98.5 ms per step, 256 96.0, 1024 97.1, 16 101.3, as dispatched 104.4)
        const int st_ = bsx_order_block(vb, nvb, H.xcd_map >= 2 ? max(2u, H.xcd_map * BSX_HSCAN_WPB / HG_WPB) : H.xcd_map, b_);
        if (st_ == 2) break;
        if (st_ == 1) continue;
        const uint32_t g = b_ * HG_WPB + (uint32_t)wv;
        if (g >= n_groups) continue;
        const uint32_t slot = rfl(H.glist[g < n_big ? g : H.task_cap - 1u - (g - n_big)]);   // the groups of HG_BIG tasks and more first: the short ones fill the tail of the launch
        const uint32_t K = rfl(H.ghead[slot]);
        // lane j < K: task j of the group
        uint32_t tid = 0, th = 0, tc0 = 0, tn = 0, key = 0, hh = 0, flags = 0, tx = 0, tw = 0;
        if ((uint32_t)lane < K) {
            tid = H.order[slot + lane];
            const HTask tk = H.tasks[tid];
            th = tk.h; tc0 = tk.c0; tn = tk.n; key = tk.key; hh = tk.sub_h; flags = tk.flags; tx = tk.pad[0]; tw = tk.pad[1];
        }
        const uint32_t n0 = rfl(tn), f0 = rfl(flags);
        if (n0 == 0) {   // slots neutralised by a refused request: their units have not published a list
            if ((uint32_t)lane < K) { HTaskOut *o = &H.tout[tid]; o->count = 0; o->overflow = 0; o->acc[0] = o->acc[1] = o->acc[2] = o->acc[3] = 0; o->c0 = 0; o->n = 0; }
        } else if (f0 & 4u) {   // RRBS (hp_task takes WGBS lists only); reads of 65-96 nt have their own code
            if (((f0 >> 8) & 15u) == 3u) hs_group<3, true, STATS>(A, H, L, lane, wv, K, tid, th, tc0, rfl(key), n0, rfl(hh), f0, rfl(tx), rfl(tw));
            else hs_group<0, true, STATS>(A, H, L, lane, wv, K, tid, th, tc0, rfl(key), n0, rfl(hh), f0, rfl(tx), rfl(tw));
        }
        else if (K > 1) {   // (a lone task as a group of one instead of through hp_task: measured slower, C3 222.4-222.8 against 218.2-218.9 ms per step, gpurun_out/r06t)
            if (((f0 >> 8) & 15u) == 5u) hs_group<5, false, STATS>(A, H, L, lane, wv, K, tid, th, tc0, rfl(key), n0, rfl(hh), f0, 0u, 0u);
            else if (((f0 >> 8) & 15u) == 4u) hs_group<4, false, STATS>(A, H, L, lane, wv, K, tid, th, tc0, rfl(key), n0, rfl(hh), f0, 0u, 0u);   // 97-128 nt (C2: 100 nt single-end)
            else hs_group<0, false, STATS>(A, H, L, lane, wv, K, tid, th, tc0, rfl(key), n0, rfl(hh), f0, 0u, 0u);
        }
        else hp_task<STATS>(A, H, rfl(tid), lane, L.W[wv].one.TAB, L.W[wv].one.PT, L.W[wv].one.Q);
        wave_fence();
    }
}

// The groups of a pass: a wave takes 64 consecutive slots of the scan order, splits their tasks into groups of up to HG_R with equal
// (first entry, length, strand copy, read offset, read words) — any of the 64, not only neighbours —, writes the slots back with
// the members of each group next to each other, in the order the groups were opened, and the group sizes into ghead.  Tasks
// that span sub-ranges are groups of one; neutralised slots (n = 0) form one group.  The start slots of the groups go into glist, a wave's
// groups next to each other, the waves' parts in the order their atomic additions arrive — nearly the scan order, which is all the
// scan kernel needs (neighbouring waves on the same cache lines); glist[cap], glist[cap + 1] count them (zeroed by k_bin_scan).
__global__ __launch_bounds__(256) void k_task_groups(const HTask *tasks, const uint32_t *n_tasks_ptr, uint32_t cap, uint32_t *order, uint32_t *ghead, uint32_t *glist)
{
    const int lane = threadIdx.x & 63;
    const uint32_t n_tasks = min(*n_tasks_ptr, cap);
    for (uint32_t s0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 64u; s0 < n_tasks; s0 += gridDim.x * 256u) {
        const uint32_t nj = min(64u, n_tasks - s0);
        uint32_t tid = 0, tn = 0, key = 0, hh = 0, flags = 0, tx = 0, tw = 0;
        if ((uint32_t)lane < nj) {
            tid = order[s0 + lane];
            const HTask tk = tasks[tid];
            tn = tk.n; key = tk.key; hh = tk.sub_h; flags = tk.flags; tx = tk.pad[0]; tw = tk.pad[1];   // (pad: the tag filter of an RRBS list, 0 otherwise)
        }
        u64 todo = bsx_ballot((uint32_t)lane < nj);
        uint32_t off = 0, head = 0, dst = 0;
        while (todo) {
            const uint32_t i0 = (uint32_t)__builtin_ctzll(todo);
            const uint32_t n0 = rl_u(tn, i0), f0 = rl_u(flags, i0), key0 = rl_u(key, i0), h0 = rl_u(hh, i0), tx0 = rl_u(tx, i0), tw0 = rl_u(tw, i0);
            const bool pending = (todo >> lane) & 1;
            u64 mem = 1ull << i0;
            if (n0 == 0) mem = bsx_ballot(pending && tn == 0);
            else if (f0 & 2u) {
                const bool same = pending && tn == n0 && key == key0 && hh == h0 && flags == f0 && tx == tx0 && tw == tw0;
                mem = bsx_ballot(same);
                if ((uint32_t)__builtin_popcountll(mem) > HG_R) mem = bsx_ballot(same && (uint32_t)__builtin_popcountll(mem & lanemask_lt(lane)) < HG_R);
            }
            const uint32_t K = (uint32_t)__builtin_popcountll(mem);
            if ((mem >> lane) & 1) dst = off + (uint32_t)__builtin_popcountll(mem & lanemask_lt(lane));
            if ((uint32_t)lane == off) head = K;
            off += K; todo &= ~mem;
        }
        if ((uint32_t)lane < nj) { order[s0 + dst] = tid; ghead[s0 + lane] = head; }
        // groups of HG_BIG tasks and more from the front of glist, the others from its back (they cannot meet: a slot starts at most one group)
        const u64 hb = bsx_ballot(head >= HG_BIG), hs = bsx_ballot(head != 0 && head < HG_BIG);
        uint32_t base = 0;
        if (lane == 0 && hb) base = atomicAdd(&glist[cap], (uint32_t)__builtin_popcountll(hb));
        if (lane == 1 && hs) base = atomicAdd(&glist[cap + 1], (uint32_t)__builtin_popcountll(hs));
        const uint32_t base_b = rl_u(base, 0), base_s = rl_u(base, 1);
        if (head >= HG_BIG) glist[base_b + (uint32_t)__builtin_popcountll(hb & lanemask_lt(lane))] = s0 + (uint32_t)lane;
        else if (head != 0) glist[cap - 1u - (base_s + (uint32_t)__builtin_popcountll(hs & lanemask_lt(lane)))] = s0 + (uint32_t)lane;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_hscan_shared — RRBS: one pass of the reference over a window of candidates for up to HS_SHARE reads at once.
// RRBS reads start at restriction sites, so the reads that fall into one repeat family walk the same bucket with the same
// read offset h (align.cpp:175-252: one seed per segment, all starts 0): after the task sort, runs of tasks that cover
// exactly the same window — same first entry, same length, same h and tag filter — sit next to each other (tens of
// thousands of reads per window on the hg38-sized genome).  A wave takes HS_SHARE consecutive tasks of the scan order, splits
// them into such runs, and per run loads each candidate's entry and reference words ONCE (funnel-shifted into the read
// frame once, too) and evaluates every read of the run against them: the per-candidate gather — the limit of the one-read
// kernel on RRBS (texture addresser busy 0.94) — is shared, only the XOR/popcount work is per read.
// Results per task are exactly those of k_hscan: survivors in list order, candidate count and word count (align.h:189-197).
// ---------------------------------------------------------------------------------------------------------------
#define HS_SHARE 16u  /* tasks per wave */
#ifndef BSX_HSHARED_W3
#define BSX_HSHARED_W3 1  /* reads of 65-96 nt through shared_window<3> */
#endif
#ifndef BSX_HSHARED_C2
#define BSX_HSHARED_C2 1  /* ... two chunks of 64 candidates per step there */
#endif
#ifndef BSX_HSHARED_WAVES
#define BSX_HSHARED_WAVES 5  /* waves per SIMD the register budget is set for */
#endif
// Plane form: the candidate's pairs of the plane copy are funnel-shifted into the READ's frame once per candidate (all reads of a run
// have the same offset, hence the same shift) and every read of the run is compared 32 nt at a time: three v_bitop3 and one v_bcnt
// per word (bsx_plane_mismatch) against four instructions per 16 nt on the packed copy.  Read frame word j holds read nt
// [32 j, 32 j + 32); the reference's first 64-bit word holds read nt [0, 32 - k), k = position mod 16: `him` on word 0, and on
// word 1 for its second early-out (align.h:189-197).
struct SharedChunk { uint32_t wd[12]; uint32_t pm1, strand; bool valid; };  // wd: pairs (pm1 >> 5) .. + 5 of the plane copy, {low, high} each

// the two halves of a chunk's fetch: the index entry, and — one step later, when the entry has arrived — the candidate's pairs of the plane copy.  (Round 6: as one function the
// entry's round trip stood in front of the gathers in every step, and the chromosome anchors behind a pointer that may be LDS or global made the gathers FLAT loads, behind
// which the compiler waits for everything: nothing of the "next chunk" was in flight while the reads were evaluated.)
__device__ __forceinline__ U2 shared_entry(const U2 *__restrict__ ent2, uint32_t idx, bool in_range)
{
    U2 e; e.a = 0xFFFFFFFFu; e.b = 0;   // (a lane behind the window's end loads nothing; shared_gather is told so again and never looks at these words)
    if (in_range) e = ent2[idx];
    return e;
}
template <int NWR, bool LDS_CHR>
__device__ __forceinline__ SharedChunk shared_gather(const U2 e, bool in_range, uint32_t h, uint32_t tag_xor, uint32_t tag_want, const uint32_t *anch_lds,
                                                     const uint32_t *__restrict__ anch_glb, const uint8_t *plane, uint32_t rc_off, int nwr)
{
    SharedChunk c;
    const uint32_t rchr = e.a & 0xffffu;
    c.valid = in_range && ((e.a ^ tag_xor) >> 16) == tag_want && e.b >= h;  // mode or strand not match / underflow the start of refseq
    uint32_t an = 0;   // (two loads of known address spaces, not one through a pointer that may be either: that is a FLAT load, and the compiler waits for every load in flight behind it)
    typedef __attribute__((address_space(3))) const uint32_t LdsU32;
    if (c.valid) an = LDS_CHR ? ((LdsU32 *)anch_lds)[rchr >> 1] : ldm1(anch_glb + (rchr >> 1));
    c.pm1 = c.valid ? an + (e.b - h) - 1u : 31u;
    c.strand = rchr & 1u;
    // one pair early when the position is pair-aligned, so that the shift into the read's frame is one v_alignbit with a 5-bit amount
    const uint32_t boff = c.valid ? hp_boff(c.pm1, c.strand ? rc_off : 0u) : 0u;
    const U4 r0 = *reinterpret_cast<const U4 *>(plane + boff);
    U4 r1, r2;
    r1.a = r1.b = r1.c = r1.d = 0; r2.a = r2.b = r2.c = r2.d = 0;
    if (c.valid && nwr > 1) r1 = *reinterpret_cast<const U4 *>(plane + boff + 16);
    if ((NWR == 0 || NWR > 3) && c.valid && nwr > 3) r2 = *reinterpret_cast<const U4 *>(plane + boff + 32);
    c.wd[0] = r0.a; c.wd[1] = r0.b; c.wd[2] = r0.c; c.wd[3] = r0.d; c.wd[4] = r1.a; c.wd[5] = r1.b; c.wd[6] = r1.c; c.wd[7] = r1.d;
    c.wd[8] = r2.a; c.wd[9] = r2.b; c.wd[10] = r2.c; c.wd[11] = r2.d;
    return c;
}

// one read of a run against one chunk of 64 candidates: the counts, the work counters' two classes, the survivors in list order behind the read's earlier ones
template <int NWR, bool PLAIN, bool STATS>
__device__ __forceinline__ void shared_eval(const HeavyArgs &H, const uint32_t *uw, uint32_t k, const uint32_t (&flo)[5], const uint32_t (&fhi)[5], uint32_t him, u64 vm, uint32_t ord,
                                            uint32_t strand, uint32_t hloc, int nwr, const uint4 &a0, const uint4 &a1, const uint4 &a2, const uint4 &a3, uint32_t thr, int lane,
                                            uint32_t &c15, uint32_t &nsv)
{
    uint32_t w0ref, w01ref, tot;
    same_counts<NWR, PLAIN, true, STATS>(flo, fhi, him, nwr, a0, a1, a2, a3, thr, vm, w0ref, w01ref, tot);
    const u64 bp = bsx_ballot(tot <= thr) & vm;
    if (STATS) {   // lane k keeps read k's counters: candidates beyond the first word | five-word candidates << 16 (work counters only)
        const u64 b1 = bsx_ballot(w0ref > thr) & vm, b5 = bsx_ballot(w01ref <= thr) & vm;
        const uint32_t add15 = (uint32_t)__builtin_popcountll(b1) | ((uint32_t)__builtin_popcountll(b5) << 16);
        if ((uint32_t)lane == k) c15 += add15;
    }
    if (bp) {
        const uint32_t base = rl_u(nsv, k);
        const uint32_t pos = base + (uint32_t)__builtin_popcountll(bp & lanemask_lt(lane));
        if (((bp >> lane) & 1) && pos < HS_SCAP) {
            SurvRec r; r.w_ord = tot | ord; r.hchr = strand; r.hloc = hloc; r.hkey = 0;
            H.tout[uw[k * 20u + 16u]].surv[pos] = r;
        }
        if ((uint32_t)lane == k) nsv += (uint32_t)__builtin_popcountll(bp);
    }
}

// one window of candidates against the K reads of a run (rows of 20 dwords at uw).  NWR = 3: reads of 65-96 nt, the RRBS headline's 75 — no fourth and fifth word, no
// third 16-byte gather, and the row in the order [X0 Y0 X1 Y1 | X2 Y2 M2 threshold | M0 M1 - -]: a read without N in its first 64 nt (bit 16 of the threshold word) takes
// two 16-byte LDS reads and two v_bitop3 per inner word (round 6: the generic form spent 7 of its 19-21 vector instructions per evaluation on the absent words —
// four zeroed registers, the materialised word-count test — and moved three row units).  NWR = 0: any length, the row in k_hscan_same's order.
template <int NWR, bool STATS, bool LDS_CHR>
__device__ __forceinline__ void shared_window(const HeavyArgs &H, const uint32_t *uw, const U2 *ent2, uint32_t n, uint32_t h, uint32_t tx, uint32_t tw,
                                              const uint32_t *anch_lds, const uint32_t *anch_glb, const uint8_t *plane, uint32_t rc_off, int nwr, uint32_t K, int lane,
                                              uint32_t &c15, uint32_t &nsv, uint32_t &nv)
{
    // C chunks of 64 candidates per step: a read's row is fetched from LDS once per step, not per chunk (the three-word form without the work counters: registers for two)
    constexpr int C = (NWR == 3 && !STATS && BSX_HSHARED_C2) ? 2 : 1;
    constexpr uint32_t STEP = 64u * C;
    SharedChunk cur[C], nxt[C];
    U2 en[C];   // the entries of the step after this one
#pragma unroll
    for (int u = 0; u < C; u++) { const uint32_t i_ = (uint32_t)(u * 64 + lane); cur[u] = shared_gather<NWR, LDS_CHR>(shared_entry(ent2, i_, i_ < n), i_ < n, h, tx, tw, anch_lds, anch_glb, plane, rc_off, nwr); }
#pragma unroll
    for (int u = 0; u < C; u++) { const uint32_t i_ = STEP + (uint32_t)(u * 64 + lane); en[u] = shared_entry(ent2, i_, i_ < n); }
    // (the first step's gathers are waited for HERE: left pending into the loop they make the compiler wait for every load in flight — the coming step's too — before each step's evaluation)
    asm volatile("" :: "v"(cur[C - 1].wd[NWR == 3 ? 7 : 11]), "v"(cur[0].wd[NWR == 3 ? 7 : 11]));
    for (uint32_t cb = 0; cb < n; cb += STEP) {
        const bool more = cb + STEP < n;
        if (more) {   // the coming step's gathers (its entries arrived during the last step) and the entries of the step behind it fly while this step's reads are evaluated
#pragma unroll
            for (int u = 0; u < C; u++) { const uint32_t i_ = cb + STEP + (uint32_t)(u * 64 + lane); nxt[u] = shared_gather<NWR, LDS_CHR>(en[u], i_ < n, h, tx, tw, anch_lds, anch_glb, plane, rc_off, nwr); }
#pragma unroll
            for (int u = 0; u < C; u++) { const uint32_t i_ = cb + 2u * STEP + (uint32_t)(u * 64 + lane); en[u] = shared_entry(ent2, i_, i_ < n); }
        }
        // the candidates' reference planes in the read frame — the same for every read of the run
        uint32_t flo[C][5], fhi[C][5], him[C], ord[C];
        u64 vm[C];
#pragma unroll
        for (int u = 0; u < C; u++) {
            const uint32_t shf = 31u - (cur[u].pm1 & 31u);                   // 32 - ((pm1 & 31) + 1)
            him[u] = 0xFFFFFFFFu << ((cur[u].pm1 + 1u) & 15u);               // read nt [0, 32 - k), k = position mod 16
#pragma unroll
            for (int t = 0; t < 5; t++) { flo[u][t] = __builtin_amdgcn_alignbit(cur[u].wd[2 * t], cur[u].wd[2 * t + 2], shf); fhi[u][t] = __builtin_amdgcn_alignbit(cur[u].wd[2 * t + 1], cur[u].wd[2 * t + 3], shf); }
            nv += cur[u].valid ? 1u : 0u;
            ord[u] = (cb + (uint32_t)(u * 64 + lane)) << 8;
            vm[u] = bsx_ballot(cur[u].valid);
        }
        const uint4 *up = reinterpret_cast<const uint4 *>(uw);
        for (uint32_t k = 0; k < K; k++, up += 5) {
            // (a 16-byte LDS read of a wave moves 1 KB — 8 cycles of the CU's LDS path, which the vector instructions of an evaluation do not hide four times over;
            //  the loop is kept lean on purpose: the same evaluations with seven more vector and a few more scalar instructions per read ran 25 % slower, r06/run27.sh)
            if (NWR == 3) {
                const uint4 u0 = up[0], u1 = up[1];   // X0 Y0 X1 Y1 | X2 Y2 M2 threshold
                const uint32_t tp = rfl(u1.w), thr = tp & 0xffffu;
                const uint4 a2 = make_uint4(u1.z, 0u, 0u, 0u), a3 = make_uint4(0u, 0u, 0u, 0u);
                if (tp >> 16) {
                    const uint4 a0 = make_uint4(u0.x, u0.y, 0xFFFFFFFFu, u0.z), a1 = make_uint4(u0.w, 0xFFFFFFFFu, u1.x, u1.y);
#pragma unroll
                    for (int u = 0; u < C; u++) {
                        if (u && !vm[u]) continue;   // (the window ends inside the step's first chunk)
                        shared_eval<NWR == 3 ? 3 : 0, NWR == 3, STATS>(H, uw, k, flo[u], fhi[u], him[u], vm[u], ord[u], cur[u].strand, cur[u].pm1 + 1u, nwr, a0, a1, a2, a3, thr, lane, c15, nsv);
                    }
                } else {
                    const uint4 u2 = up[2];           // M0 M1 - -
                    const uint4 a0 = make_uint4(u0.x, u0.y, u2.x, u0.z), a1 = make_uint4(u0.w, u2.y, u1.x, u1.y);
#pragma unroll
                    for (int u = 0; u < C; u++) {
                        if (u && !vm[u]) continue;
                        shared_eval<NWR == 3 ? 3 : 0, false, STATS>(H, uw, k, flo[u], fhi[u], him[u], vm[u], ord[u], cur[u].strand, cur[u].pm1 + 1u, nwr, a0, a1, a2, a3, thr, lane, c15, nsv);
                    }
                }
            } else {
                const uint4 a0 = up[0], a1 = up[1], a2 = up[2];  // X0 Y0 M0 X1 | Y1 M1 X2 Y2 | M2 threshold X3 Y3 | M3 X4 Y4 M4
                uint4 a3 = make_uint4(0u, 0u, 0u, 0u);
                if (nwr > 3) a3 = up[3];
                const uint32_t thr = rfl(a2.y) & 0xffffu;
                shared_eval<0, false, STATS>(H, uw, k, flo[0], fhi[0], him[0], vm[0], ord[0], cur[0].strand, cur[0].pm1 + 1u, nwr, a0, a1, a2, a3, thr, lane, c15, nsv);
            }
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < C; u++) cur[u] = nxt[u];
        }
    }
}

template <bool STATS>
__global__ __launch_bounds__(256, BSX_HSHARED_WAVES) void k_hscan_shared(AlignArgs A, HeavyArgs H)
{
    __shared__ __attribute__((aligned(16))) uint32_t UW[4][HS_SHARE][20];   // per read of the run: X0 Y0 M0 X1 | Y1 M1 X2 Y2 | M2 threshold X3 Y3 | M3 X4 Y4 M4 | task id
    __shared__ uint32_t ANCH[BSX_LDS_CHR + 1];
    const DevParams &P = A.P;
    const int lane = threadIdx.x & 63, wv = (int)rfl(threadIdx.x >> 6);  // (the wave number as a scalar: what depends on it stays wave-uniform for the compiler)
    if (P.n_chr <= BSX_LDS_CHR) for (uint32_t i = threadIdx.x; i <= P.n_chr; i += 256) ANCH[i] = P.anchor[i];
    __syncthreads();
    const bool lds_chr = P.n_chr <= BSX_LDS_CHR;
    const uint32_t n_tasks = min(*H.n_tasks, H.task_cap);
    const uint8_t *plane = reinterpret_cast<const uint8_t *>(P.refplane);
    const uint32_t rc_off = P.plane_rc_off;
    // (grid sized for the task pool, or — in the tail of a batch — smaller: then a wave sweeps over the order with the stride of the grid)
    const uint32_t nvb = (n_tasks + 4u * HS_SHARE - 1u) / (4u * HS_SHARE);
    for (uint32_t vb = blockIdx.x;; vb += gridDim.x) {
    uint32_t b_;
    const int st_ = bsx_order_block(vb, nvb, H.xcd_map >= 2 ? max(2u, H.xcd_map / 32u) : H.xcd_map, b_);   // (a block takes 64 tasks: 32 of k_hscan's blocks)
    if (st_ == 2) break;
    const uint32_t s0 = (b_ * 4u + (uint32_t)wv) * HS_SHARE;
    if (st_ == 1 || s0 >= n_tasks) continue;
    const uint32_t nj = min(HS_SHARE, n_tasks - s0);
    // lane j < nj: task j of this wave, in scan order, and the signature of its window
    uint32_t tid = 0, th = 0, tc0 = 0, tn = 0, tkey = 0, sh_ = 0, stx = 0, stw = 0, snw = 0;
    if ((uint32_t)lane < nj) {
        tid = H.order ? H.order[s0 + lane] : s0 + (uint32_t)lane;
        const HTask tk = H.tasks[tid];
        th = tk.h; tc0 = tk.c0; tn = tk.n; tkey = tk.key;
        if (tn) {
            const ListReq &R = H.state[th & 0x3fffffffu].req[th >> 30];
            tkey = R.sub_base[0] + (tc0 - R.sub_pre[0]);  // first entry of the window
            sh_ = R.sub_h[0]; stx = R.tag_xor; stw = R.tag_want; snw = (R.len + 31u) >> 5;  // snw: read words of 32 nt
        } else {  // slot neutralised by a refused request: its unit has not published a list
            HTaskOut *o = &H.tout[tid];
            o->count = 0; o->overflow = 0; o->acc[0] = o->acc[1] = o->acc[2] = o->acc[3] = 0; o->c0 = 0; o->n = 0;
        }
    }
    u64 st_cand = 0, st_words = 0, st_n1 = 0, st_n5 = 0;  // statistics of the scan kernel (lane 0)
    for (uint32_t i0 = 0; i0 < nj;) {
        if (rl(tn, (int)i0) == 0) { i0++; continue; }
        // the run of tasks from i0 that cover exactly the same window
        const bool same = (uint32_t)lane >= i0 && (uint32_t)lane < nj && tn == rl(tn, (int)i0) && tkey == rl(tkey, (int)i0) && sh_ == rl(sh_, (int)i0) &&
                          stx == rl(stx, (int)i0) && stw == rl(stw, (int)i0) && snw == rl(snw, (int)i0);
        const u64 sm = bsx_ballot(same) >> i0;
        const uint32_t K = (uint32_t)__builtin_ctzll(~sm);  // (bit 0 is set: the task equals itself)
        const uint32_t key = rl(tkey, (int)i0), n = rl(tn, (int)i0), h = rl(sh_, (int)i0), tx = rl(stx, (int)i0), tw = rl(stw, (int)i0);
        const int nwr = (int)rl(snw, (int)i0);
        wave_fence();
        for (uint32_t xb = 0; xb < K * 20u; xb += 64) {
            const uint32_t x = xb + (uint32_t)lane, k = min(x / 20u, K - 1u), f = x - k * 20u;
            const uint32_t hk = (uint32_t)__shfl((int)th, (int)(i0 + k)), tk = (uint32_t)__shfl((int)tid, (int)(i0 + k));  // (all lanes take part)
            if (x < K * 20u) {
                const ListReq &R = H.state[hk & 0x3fffffffu].req[hk >> 30];
                uint32_t v = 0;
                const uint32_t j = f / 3u, c = f - 3u * j;
                if (f < 15) v = c == 0 ? R.px[j] : c == 1 ? R.py[j] : R.pm[j];
                else if (f == 15) v = R.thres | ((BSX_HSHARED_W3 && nwr == 3 && (R.pm[0] & R.pm[1]) == 0xFFFFFFFFu) ? 0x10000u : 0u);   // bit 16: no N in the first 64 nt (shared_window<3>)
                else if (f == 16) v = tk;
                uint32_t at = f < 9 ? f : f < 15 ? f + 1u : f == 15 ? 9u : f;   // the threshold behind word 2: reads of up to 96 nt need three of the row's 16-byte units
                if (BSX_HSHARED_W3 && nwr == 3)   // X0 Y0 X1 Y1 | X2 Y2 M2 threshold | M0 M1 - - | (words 3, 4: absent) | task
                    at = f >= 15 ? (f == 15 ? 7u : f) : j > 2 ? 1u + f : c == 2 ? (j == 2 ? 6u : 8u + j) : (j == 2 ? 4u + c : 2u * j + c);
                UW[wv][k][at] = v;
            }
        }
        uint32_t c15 = 0, nsv = 0;  // lane k: counters of read k (see the loop)
        wave_fence();
        const U2 *ent2 = reinterpret_cast<const U2 *>(P.entries) + key;
        uint32_t nv = 0;
        if (lds_chr) {
            if (BSX_HSHARED_W3 && nwr == 3) shared_window<3, STATS, true>(H, &UW[wv][0][0], ent2, n, h, tx, tw, ANCH, P.anchor, plane, rc_off, nwr, K, lane, c15, nsv, nv);
            else shared_window<0, STATS, true>(H, &UW[wv][0][0], ent2, n, h, tx, tw, ANCH, P.anchor, plane, rc_off, nwr, K, lane, c15, nsv, nv);
        } else {   // (more chromosomes than the LDS table holds: the anchors from global memory)
            if (BSX_HSHARED_W3 && nwr == 3) shared_window<3, STATS, false>(H, &UW[wv][0][0], ent2, n, h, tx, tw, ANCH, P.anchor, plane, rc_off, nwr, K, lane, c15, nsv, nv);
            else shared_window<0, STATS, false>(H, &UW[wv][0][0], ent2, n, h, tx, tw, ANCH, P.anchor, plane, rc_off, nwr, K, lane, c15, nsv, nv);
        }
        wave_fence();
        const uint32_t n_cand = wave_sum(nv);
        {
            const bool mine = (uint32_t)lane < K;
            const uint32_t my_c0 = (uint32_t)__shfl((int)tc0, (int)min(i0 + (uint32_t)lane, 63u));  // (all lanes take part) the list ordinal read `lane`'s task starts at
            const uint32_t ns = nsv, n1 = c15 & 0xffffu, n5 = c15 >> 16;
            const bool ov = ns > HS_SCAP;
            if (mine) {
                HTaskOut *o = &H.tout[UW[wv][lane][16]];
                o->count = min(ns, (uint32_t)HS_SCAP); o->overflow = ov ? 1 : 0; o->acc[0] = n_cand; o->acc[1] = STATS ? 2u * n_cand - n1 + 3u * n5 : 0u; o->acc[2] = 0; o->acc[3] = 0; o->c0 = my_c0; o->n = n;
            }
            // work of the scan kernel (incl. speculation); an overflowed task is redone by the control kernel
            const bool cnt = mine && !ov;
            const uint32_t kk = (uint32_t)__builtin_popcountll(bsx_ballot(cnt));
            const uint32_t s1 = wave_sum(cnt ? n1 : 0), s5 = wave_sum(cnt ? n5 : 0);
            st_cand += (u64)kk * n_cand;
            if (STATS) { st_n1 += s1; st_n5 += s5; st_words += 2ull * kk * n_cand - s1 + 3ull * s5; }
        }
        wave_fence();
        i0 += K;
    }
    if (lane == 0) {
        u64 *sh = (u64 *)A.scan_stats + (size_t)(blockIdx.x & 63u) * 8;
        atomicAdd((u64 *)&sh[0], st_cand); atomicAdd((u64 *)&sh[1], st_words); atomicAdd((u64 *)&sh[2], st_n1); atomicAdd((u64 *)&sh[3], st_n5);
    }
    wave_fence();
    }
}
}  // namespace

// exact mode pre-pass (see k_leak_meta): with_meta = the stream's records are not up to date (new reads / history); `final_out` != null:
// also leave the state behind the stream's last read there
void bsx_launch_leak(const AlignArgs &A, int paired, int n_cu, bool with_meta, bool resolve, void *final_out, hipStream_t stream)
{
    const uint32_t n_stream = A.n_hist + A.n_units_all;
    if (with_meta) {
        const int g = (int)std::max<uint32_t>(1, std::min<uint32_t>((n_stream + 3) / 4, (uint32_t)n_cu * 8));
        if (paired) hipLaunchKernelGGL(k_leak_meta<true>, dim3(g), dim3(256), 0, stream, A);
        else hipLaunchKernelGGL(k_leak_meta<false>, dim3(g), dim3(256), 0, stream, A);
    }
    if (resolve) {
        const int g = (int)std::max<uint32_t>(1, std::min<uint32_t>((A.n_units - A.first_unit + 3) / 4, (uint32_t)n_cu * 8));
        if (paired) hipLaunchKernelGGL(k_leak_resolve<true>, dim3(g), dim3(256), 0, stream, A);
        else hipLaunchKernelGGL(k_leak_resolve<false>, dim3(g), dim3(256), 0, stream, A);
    }
    if (final_out) {
        if (paired) hipLaunchKernelGGL(k_leak_final<true>, dim3(2), dim3(64), 0, stream, A, (LeakState *)final_out);
        else hipLaunchKernelGGL(k_leak_final<false>, dim3(2), dim3(64), 0, stream, A, (LeakState *)final_out);
    }
}
size_t bsx_leakrec_bytes(void) { return sizeof(LeakRec); }
size_t bsx_leakstate_bytes(void) { return sizeof(LeakState); }
uint32_t bsx_leak_blk(void) { return LEAK_BLK; }

void bsx_launch_align(const AlignArgs &A, int paired, int grid_blocks, hipStream_t stream)
{
    const bool ctx = A.P.ctx && !A.work_counters && A.P.index_interval <= 4 && !A.P.rrbs;
    if (A.leak_exact && ctx) {   // (round 6: the exact mode keeps the context prefilter — it used to fall back to the plain scan, most of its 11 % on C5)
        if (paired) hipLaunchKernelGGL((k_align<true, true, true>), dim3(grid_blocks), dim3(256), 0, stream, A);
        else hipLaunchKernelGGL((k_align<false, true, true>), dim3(grid_blocks), dim3(256), 0, stream, A);
    } else if (A.leak_exact) {
        if (paired) hipLaunchKernelGGL((k_align<true, true>), dim3(grid_blocks), dim3(256), 0, stream, A);
        else hipLaunchKernelGGL((k_align<false, true>), dim3(grid_blocks), dim3(256), 0, stream, A);
    } else if (A.P.ctx && !A.work_counters && A.P.index_interval <= 4 && !A.P.rrbs) {   // the context prefilter: with the index's flank words (the kernel's flank table holds four phases: -I <= 4, WGBS), and only where nobody reads the work counters
        if (paired) hipLaunchKernelGGL((k_align<true, false, true>), dim3(grid_blocks), dim3(256), 0, stream, A);
        else hipLaunchKernelGGL((k_align<false, false, true>), dim3(grid_blocks), dim3(256), 0, stream, A);
    } else if (paired) hipLaunchKernelGGL((k_align<true, false>), dim3(grid_blocks), dim3(256), 0, stream, A);
    else hipLaunchKernelGGL((k_align<false, false>), dim3(grid_blocks), dim3(256), 0, stream, A);
}

void bsx_launch_hctrl(const AlignArgs &A, const HeavyArgsRaw &R, int paired, int grid_blocks, hipStream_t stream)
{
    const HeavyArgs H = typed(R);
    if (paired) hipLaunchKernelGGL(k_hctrl<true>, dim3(grid_blocks), dim3(256), 0, stream, A, H);
    else hipLaunchKernelGGL(k_hctrl<false>, dim3(grid_blocks), dim3(256), 0, stream, A, H);
}

// The scan launches are sized for the group's whole task pool: the number of tasks a control pass published stays on the device
// (H.n_tasks), blocks beyond it exit at once (28 us for 131 072 empty blocks, profiles/r03_launch_cost.json) — no host read-back
// between a control pass and its scan.
// max_tasks: 0 = a grid for the whole task pool; otherwise a grid for that many tasks, whose blocks sweep over whatever the pass published
void bsx_launch_hscan_shared(const AlignArgs &A, const HeavyArgsRaw &R, hipStream_t stream, uint32_t max_tasks)
{
    const HeavyArgs H = typed(R);
    const uint32_t jobs = ((max_tasks ? std::min(max_tasks, R.task_cap) : R.task_cap) + HS_SHARE - 1) / HS_SHARE;
    // (a multiple of 8: order_block)
    if (A.work_counters) hipLaunchKernelGGL(k_hscan_shared<true>, dim3(((jobs + 3) / 4 + 7u) & ~7u), dim3(256), 0, stream, A, H);
    else hipLaunchKernelGGL(k_hscan_shared<false>, dim3(((jobs + 3) / 4 + 7u) & ~7u), dim3(256), 0, stream, A, H);
}

void bsx_launch_hscan_same(const AlignArgs &A, const HeavyArgsRaw &R, hipStream_t stream, uint32_t max_tasks)
{
    const HeavyArgs H = typed(R);
    // The host does not know the number of groups: the grid is sized for a quarter of the task pool (a group holds 4-5 tasks on average; where a pass
    // has more groups the blocks sweep) — surplus blocks leave at once, but the dispatcher still has to start them: a grid for the whole pool 98.8 and
    // 144.7 ms per step (WGBS, RRBS), a quarter 96.2 and 142.2, a sixteenth 100.3 and 142.0, 1/64 (RRBS) 150.2
    static const uint32_t grid_div = getenv("BSX_SAME_GRID_DIV") ? (uint32_t)std::max(1, atoi(getenv("BSX_SAME_GRID_DIV"))) : 4u;
    uint32_t blocks = ((max_tasks ? std::min(max_tasks, R.task_cap) : R.task_cap / grid_div) + HG_WPB - 1) / HG_WPB;
    blocks = (blocks + 7u) & ~7u;  // the same number of blocks for each of the 8 XCDs (the sweep relies on a multiple of 8)
    if (A.work_counters) hipLaunchKernelGGL(k_hscan_same<true>, dim3(blocks), dim3(64 * HG_WPB), 0, stream, A, H);
    else hipLaunchKernelGGL(k_hscan_same<false>, dim3(blocks), dim3(64 * HG_WPB), 0, stream, A, H);
}

void bsx_launch_hscan(const AlignArgs &A, const HeavyArgsRaw &R, hipStream_t stream, uint32_t max_tasks)
{
    const HeavyArgs H = typed(R);
    uint32_t blocks = ((max_tasks ? std::min(max_tasks, R.task_cap) : R.task_cap) + BSX_HSCAN_WPB - 1) / BSX_HSCAN_WPB;
    blocks = (blocks + 7u) & ~7u;  // the same number of blocks for each of the 8 XCDs (k_hscan's sweep relies on a multiple of 8)
    if (A.work_counters) hipLaunchKernelGGL(k_hscan<true>, dim3(blocks), dim3(64 * BSX_HSCAN_WPB), 0, stream, A, H);
    else hipLaunchKernelGGL(k_hscan<false>, dim3(blocks), dim3(64 * BSX_HSCAN_WPB), 0, stream, A, H);
}

// ---------------------------------------------------------------------------------------------------------------
// scan order of a pass: task ids ordered by the index entry their window starts at, at a granularity of 2^shift entries
// (a counting sort over bins; the count of tasks stays on the device).  Tasks that walk the same part of a big bucket then
// run together and share its cache lines; exact order inside a bin does not matter for that, and results
// never depend on the order (every task writes its own record).
//   k_task_bins   : rank[i] = arrival number of task i in its bin; also zeroes the count block the NEXT control pass writes
//   k_bin_scan    : per chunk of BIN_CHUNK bins an exclusive prefix (counts -> starts, counts zeroed for the next pass) and the chunk total
//   k_task_order  : order[chunk_start + start + rank] = i
// ---------------------------------------------------------------------------------------------------------------
namespace {
#define BIN_CHUNK 2048u
// Thousands of reads walk the same window of a giant bucket: their tasks fall into ONE bin, and one memory word takes only ~88 atomics
// per microsecond (300-400 us per pass when every task went to the counter itself).  A block first counts its 1024 tasks per bin in
// an LDS table and then adds each bin's count to memory once.
#define BIN_LDS 2048u
// the bin of a task.  spread: a task that lies inside one sub-range (the pieces of a giant bucket) owns the bins its entries cover; the
// tasks of all the reads that walk that piece are dealt over those bins (up to 16) by their read offset h, so that the tasks k_hscan_same
// can evaluate together — same window, same offset — arrive next to each other, while the order by index entry is kept.
__device__ __forceinline__ uint32_t task_bin(const HTask &tk, uint32_t shift, uint32_t n_bins, uint32_t spread)
{
    uint32_t bin = tk.key >> shift;
    if (spread && (tk.flags & 2u)) {
        const uint32_t span = tk.n >> shift;
        const uint32_t nb = span >= 16u ? 16u : span >= 8u ? 8u : span >= 4u ? 4u : span >= 2u ? 2u : 1u;
        bin += ((tk.sub_h * 0x9E3779B1u) >> 28) & (nb - 1u);
    }
    return min(bin, n_bins - 1u);
}
__global__ __launch_bounds__(256) void k_task_bins(const HTask *tasks, const uint32_t *n_tasks_ptr, uint32_t cap, uint32_t shift, uint32_t n_bins, uint32_t spread, uint32_t *bins,
                                                    uint32_t *rank, uint32_t *zero_blk)
{
    __shared__ uint32_t hkey[BIN_LDS], hcnt[BIN_LDS];
    if (zero_blk && blockIdx.x == 0 && threadIdx.x < 3) zero_blk[threadIdx.x == 0 ? 0u : threadIdx.x == 1 ? BSX_HCNT_TASKS : BSX_HCNT_QUEUE] = 0;   // (a counter block: bsx_kernel_args.h)
    const uint32_t n = min(*n_tasks_ptr, cap);
    for (uint32_t chunk = blockIdx.x * 1024u; chunk < n; chunk += gridDim.x * 1024u) {
        for (uint32_t i = threadIdx.x; i < BIN_LDS; i += 256) { hkey[i] = 0xffffffffu; hcnt[i] = 0; }
        __syncthreads();
        uint32_t sl[4], lr[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t i = chunk + (uint32_t)k * 256u + threadIdx.x;
            sl[k] = 0; lr[k] = 0;
            if (i < n) {
                const uint32_t bin = task_bin(tasks[i], shift, n_bins, spread);
                uint32_t slot = (bin * 0x9E3779B1u) >> 21;  // 11 bits
                for (;;) {
                    const uint32_t prev = atomicCAS(&hkey[slot], 0xffffffffu, bin);
                    if (prev == 0xffffffffu || prev == bin) break;
                    slot = (slot + 1) & (BIN_LDS - 1);
                }
                sl[k] = slot; lr[k] = atomicAdd(&hcnt[slot], 1u);
            }
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < BIN_LDS; i += 256)
            if (hkey[i] != 0xffffffffu) hcnt[i] = atomicAdd(&bins[hkey[i]], hcnt[i]);  // count -> first rank of this block's tasks in the bin
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t i = chunk + (uint32_t)k * 256u + threadIdx.x;
            if (i < n) rank[i] = hcnt[sl[k]] + lr[k];
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_bin_scan(const uint32_t *n_tasks_ptr, uint32_t *bins, uint32_t *bstart, uint32_t *chunk_tot, uint32_t n_bins, uint32_t *zero_word)
{
    __shared__ uint32_t part[256];
    if (zero_word && blockIdx.x == 0 && threadIdx.x < 2) zero_word[threadIdx.x] = 0;   // the group counts of this pass (k_task_groups)
    if (*n_tasks_ptr == 0) return;  // (all counts are zero and stay zero; nothing reads the starts)
    const uint32_t base = blockIdx.x * BIN_CHUNK + threadIdx.x * 8u;
    uint32_t v[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { v[k] = base + k < n_bins ? bins[base + k] : 0u; sum += v[k]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < 256; o <<= 1) {
        const uint32_t t = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;  // exclusive prefix of this thread's eight bins inside the chunk
#pragma unroll
    for (int k = 0; k < 8; k++) { if (base + k < n_bins) { bstart[base + k] = run; if (v[k]) bins[base + k] = 0; } run += v[k]; }
    if (threadIdx.x == 255) chunk_tot[blockIdx.x] = part[255];
}
__global__ __launch_bounds__(256) void k_task_order(const HTask *tasks, const uint32_t *n_tasks_ptr, uint32_t cap, uint32_t shift, uint32_t n_bins, uint32_t spread, const uint32_t *bstart,
                                                     const uint32_t *chunk_tot, uint32_t n_chunks, const uint32_t *rank, uint32_t *order)
{
    __shared__ uint32_t cstart[1024];
    const uint32_t n = min(*n_tasks_ptr, cap);
    if (n == 0) return;
    // exclusive prefix of the chunk totals (n_chunks <= 1024), redone by every block: cheaper than one more launch
    for (uint32_t i = threadIdx.x; i < 1024; i += 256) cstart[i] = i < n_chunks ? chunk_tot[i] : 0u;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        uint32_t t[4];
        for (int k = 0; k < 4; k++) { const uint32_t i = threadIdx.x + 256 * k; t[k] = i >= o ? cstart[i - o] : 0u; }
        __syncthreads();
        for (int k = 0; k < 4; k++) cstart[threadIdx.x + 256 * k] += t[k];
        __syncthreads();
    }
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t b = task_bin(tasks[i], shift, n_bins, spread), c = b / BIN_CHUNK;
        order[(c ? cstart[c - 1] : 0u) + bstart[b] + rank[i]] = i;
    }
}
}  // namespace

uint32_t bsx_bin_chunks(uint32_t n_bins) { return (n_bins + BIN_CHUNK - 1) / BIN_CHUNK; }

// bins: [n_bins] zero on entry and on exit; bstart: [n_bins]; chunk_tot: [bsx_bin_chunks(n_bins)] (<= 1024 chunks); rank, order: [task_cap];
// zero_blk: the counter block to clear for the coming pass (or null)
void bsx_launch_task_order(const HeavyArgsRaw &R, uint32_t shift, uint32_t n_bins, uint32_t *bins, uint32_t *bstart, uint32_t *chunk_tot, uint32_t *rank, uint32_t *order,
                           uint32_t *zero_blk, hipStream_t stream, uint32_t spread, bool groups)
{
    const uint32_t grid = std::max(1u, std::min(512u, (R.task_cap + 255u) / 256u)), n_chunks = bsx_bin_chunks(n_bins);
    hipLaunchKernelGGL(k_task_bins, dim3(std::max(1u, std::min(256u, (R.task_cap + 1023u) / 1024u))), dim3(256), 0, stream, (const HTask *)R.tasks, R.n_tasks, R.task_cap, shift, n_bins, spread, bins, rank, zero_blk);
    hipLaunchKernelGGL(k_bin_scan, dim3(n_chunks), dim3(256), 0, stream, R.n_tasks, bins, bstart, chunk_tot, n_bins, groups ? R.glist + R.task_cap : (uint32_t *)nullptr);
    hipLaunchKernelGGL(k_task_order, dim3(grid), dim3(256), 0, stream, (const HTask *)R.tasks, R.n_tasks, R.task_cap, shift, n_bins, spread, bstart, chunk_tot, n_chunks, rank, order);
    // (the ranks are spent: their array takes the group sizes)
    if (groups) hipLaunchKernelGGL(k_task_groups, dim3(std::max(1u, std::min(1024u, (R.task_cap + 255u) / 256u))), dim3(256), 0, stream, (const HTask *)R.tasks, R.n_tasks, R.task_cap, order, rank, R.glist);
}

// ---------------------------------------------------------------------------------------------------------------
// Diagnostics (BSX_SIGHIST=1, bsx_api.hip): per pass, how many tasks cover exactly the same window (first entry, length, strand copy)
// with the same read offset h — the tasks that could share one fetch AND one shift of the candidates' reference.  Candidates are
// counted by the size R of their task's group: hist[k] for R in (2^(k-1), 2^k], hist[31] for tasks that span sub-ranges.
// ---------------------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ unsigned long long sig_of(const HTask &tk, bool with_h)
{
    unsigned long long x = ((unsigned long long)tk.key << 32) ^ ((unsigned long long)tk.n << 9) ^ (tk.flags & 1u) ^ (with_h ? (unsigned long long)tk.sub_h * 0x9E3779B97F4A7C15ull : 0ull);
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    return x | 1ull;
}
__global__ __launch_bounds__(256) void k_sig_count(const HTask *tasks, const uint32_t *n_tasks_ptr, uint32_t cap, unsigned long long *tab, uint32_t mask, int with_h)
{
    const uint32_t n = min(*n_tasks_ptr, cap);
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const HTask tk = tasks[i];
        if (!tk.n || !(tk.flags & 2u)) continue;
        const unsigned long long sg = sig_of(tk, with_h != 0);
        for (uint32_t slot = (uint32_t)(sg >> 20) & mask;; slot = (slot + 1) & mask) {
            const unsigned long long prev = atomicCAS(&tab[2 * (size_t)slot], 0ull, sg);
            if (prev == 0ull || prev == sg) { atomicAdd(&tab[2 * (size_t)slot + 1], 1ull); break; }
        }
    }
}
__global__ __launch_bounds__(256) void k_sig_hist(const HTask *tasks, const uint32_t *n_tasks_ptr, uint32_t cap, const unsigned long long *tab, uint32_t mask, int with_h, unsigned long long *hist)
{
    const uint32_t n = min(*n_tasks_ptr, cap);
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const HTask tk = tasks[i];
        if (!tk.n) continue;
        uint32_t k = 31;
        if (tk.flags & 2u) {
            const unsigned long long sg = sig_of(tk, with_h != 0);
            // (bounded probe: the table is process-global — with a second batch in flight its memset can clear the table under this kernel;
            //  the histogram is then wrong, diagnostics are meant for one batch in flight, but nothing may spin)
            uint32_t slot = (uint32_t)(sg >> 20) & mask, probes = 0;
            while (tab[2 * (size_t)slot] != sg && probes <= mask) { slot = (slot + 1) & mask; probes++; }
            if (probes <= mask) {
                const unsigned long long r = tab[2 * (size_t)slot + 1];
                k = r <= 1 ? 0u : 32u - (uint32_t)__builtin_clz((uint32_t)r - 1u);
            }
        }
        atomicAdd(&hist[k], (unsigned long long)tk.n);
        atomicAdd(&hist[32 + k], 1ull);
    }
}
unsigned long long *g_sig_tab = nullptr, *g_sig_hist = nullptr;
const uint32_t SIG_SLOTS = 1u << 23;
}  // namespace

void bsx_sig_hist_pass(const HeavyArgsRaw &R, hipStream_t stream)
{
    if (!g_sig_tab) {
        if (hipMalloc((void **)&g_sig_tab, (size_t)SIG_SLOTS * 16) != hipSuccess || hipMalloc((void **)&g_sig_hist, 2 * 64 * 8) != hipSuccess) { g_sig_tab = nullptr; return; }
        (void)hipMemsetAsync(g_sig_hist, 0, 2 * 64 * 8, stream);
    }
    for (int with_h = 0; with_h < 2; with_h++) {
        (void)hipMemsetAsync(g_sig_tab, 0, (size_t)SIG_SLOTS * 16, stream);
        hipLaunchKernelGGL(k_sig_count, dim3(1024), dim3(256), 0, stream, (const HTask *)R.tasks, R.n_tasks, R.task_cap, g_sig_tab, SIG_SLOTS - 1, with_h);
        hipLaunchKernelGGL(k_sig_hist, dim3(1024), dim3(256), 0, stream, (const HTask *)R.tasks, R.n_tasks, R.task_cap, g_sig_tab, SIG_SLOTS - 1, with_h, g_sig_hist + 64 * with_h);
    }
}

void bsx_sig_hist_report(void)
{
    if (!g_sig_tab) return;
    unsigned long long h[128];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, g_sig_hist, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return;
    for (int with_h = 0; with_h < 2; with_h++) {
        unsigned long long tot = 0, tt = 0;
        for (int k = 0; k < 32; k++) { tot += h[64 * with_h + k]; tt += h[64 * with_h + 32 + k]; }
        fprintf(stderr, "[sighist] same window%s: candidates %llu tasks %llu\n", with_h ? " and offset" : "", tot, tt);
        for (int k = 0; k < 32; k++)
            if (h[64 * with_h + 32 + k])
                fprintf(stderr, "[sighist]   %s %-6u cand %.4f tasks %.4f\n", k == 31 ? "spanning" : "R <=", k == 31 ? 0u : 1u << k, (double)h[64 * with_h + k] / (double)std::max(1ull, tot), (double)h[64 * with_h + 32 + k] / (double)std::max(1ull, tt));
    }
    (void)hipMemset(g_sig_hist, 0, 2 * 64 * 8);
}

// BSX_SECTOR_STATS (diagnostic build only; a no-op in the shipped library): count and clear the sectors the scan launch just marked
void bsx_sector_pass(const bsx_ref *r, hipStream_t stream)
{
#ifdef BSX_SECTOR_STATS
    const uint32_t n_words = (uint32_t)(((uint64_t)r->plane_rc_off * 2 / 64 + 31) / 32 + 64);
    static uint32_t *bits = nullptr;
    if (!bits) {
        if (hipMalloc((void **)&bits, (size_t)n_words * 4) != hipSuccess) return;
        (void)hipMemset(bits, 0, (size_t)n_words * 4);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sector_bits), &bits, sizeof(bits));
        return;   // (the first call only sets the bitmap up: called once before the first scan)
    }
    hipLaunchKernelGGL(k_sector_pop, dim3(1024), dim3(256), 0, stream, n_words);
#else
    (void)r; (void)stream;
#endif
}
void bsx_sector_report(void)
{
#ifdef BSX_SECTOR_STATS
    unsigned long long t = 0, d = 0, p = 0;
    if (hipDeviceSynchronize() != hipSuccess) return;
    (void)hipMemcpyFromSymbol(&t, HIP_SYMBOL(g_sector_touch), 8); (void)hipMemcpyFromSymbol(&d, HIP_SYMBOL(g_sector_distinct), 8); (void)hipMemcpyFromSymbol(&p, HIP_SYMBOL(g_sector_passes), 8);
    fprintf(stderr, "[sectors] group scan: %llu lane-sector touches, %llu distinct sectors summed over %llu scan launches (compulsory), ratio %.2f\n", t, d, p, d ? (double)t / (double)d : 0.0);
    const unsigned long long z = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sector_touch), &z, 8); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sector_distinct), &z, 8); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sector_passes), &z, 8);
#endif
}

size_t bsx_hstate_bytes(void) { return sizeof(HState); }
size_t bsx_htask_bytes(void) { return sizeof(HTask); }
size_t bsx_htaskout_bytes(void) { return sizeof(HTaskOut); }

int bsx_align_occupancy(int paired)
{
    int nb = 0;
    hipError_t e = paired ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_align<true, false>, 256, 0)
                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_align<false, false>, 256, 0);
    if (e != hipSuccess || nb < 1) nb = 2;
    return nb;
}

int bsx_hctrl_occupancy(int paired)
{
    int nb = 0;
    hipError_t e = paired ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_hctrl<true>, 256, 0) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_hctrl<false>, 256, 0);
    if (e != hipSuccess || nb < 1) nb = 1;
    return nb;
}
