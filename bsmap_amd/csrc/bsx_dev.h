// bsx_dev.h — bit primitives shared by host and device code of libbsx.so.
//
// Reference definitions (BSMAP v2.6): Param::XT param.h:123 + BuildMismatchTable param.cpp:122-137 (3-letter hash),
// Param::XC param.h:125 (T->C mask), Param::XM param.h:129-137 (count of non-zero 2-bit groups).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// 3-letter seed hash: per nt collapse code 3 (read nucleotide, 'T') onto code 1 ('C'), then read the 16 two-bit
// digits as a base-3 number, first nt most significant.  The reference uses a 64K-entry table per 8 nt; on the GPU
// the digits are folded pairwise in four multiply-adds, no table and no memory traffic.
__host__ __device__ inline uint32_t bsx_seed_hash(uint32_t x)
{
    uint32_t d = ((~((x << 1) & x)) | 0x55555555u) & x;       // 11 -> 01 in every 2-bit group
    d = ((d >> 2) & 0x33333333u) * 3u + (d & 0x33333333u);    // 8 base-9 digits
    d = ((d >> 4) & 0x0F0F0F0Fu) * 9u + (d & 0x0F0F0F0Fu);    // 4 base-81 digits
    d = ((d >> 8) & 0x00FF00FFu) * 81u + (d & 0x00FF00FFu);   // 2 base-6561 digits
    return (d >> 16) * 6561u + (d & 0xFFFFu);
}

// per-nt mismatch bits of a read word against a reference word: bit 2i set iff nt i mismatches under the rule
// ((read & XC(ref)) ^ ref) & mask, where XC gives 01 at reference code 1 ('C') and 11 elsewhere (reference align.h:189)
__host__ __device__ inline uint32_t bsx_mismatch_bits(uint32_t read, uint32_t mask, uint32_t ref)
{
    const uint32_t xc = ((~ref) << 1) | ref | 0x55555555u;
    const uint32_t x = ((read & xc) ^ ref) & mask;
    return (x | (x >> 1)) & 0x55555555u;
}

// the same rule with the reference-dependent mask moved to the read side, where it is wave-uniform: a nt mismatches iff
// the low bit of read^ref is set, or the high bit is set and the read nt is not 'T' (read T over reference C gives 10).
// tmask = read mask with the high bit cleared at the read's T positions (bsx_tmask); result bit 2i+1 set iff nt i mismatches.
__host__ __device__ inline uint32_t bsx_tmask(uint32_t read, uint32_t mask) { return mask & ~(read & (read << 1) & 0xAAAAAAAAu); }
__host__ __device__ inline uint32_t bsx_mismatch_hi(uint32_t read, uint32_t tmask, uint32_t ref)
{
    const uint32_t y = (read ^ ref) & tmask;
    return ((y << 1) | y) & 0xAAAAAAAAu;
}

// Bit planes.  A packed word holds 16 nt, 2 bits each, first nt in the top bits (dbseq.cpp:58-111); a PLANE word holds one bit of
// 32 consecutive nt, first nt in bit 31.  bsx_plane_half: the low (which = 0) or high (which = 1) bit of the 16 nt of a packed word,
// first nt in bit 15.  The scan kernels compare 32 nt per word pair of the plane copy of the reference: the mismatch rule above
// on planes is  ((Rlo ^ X) & M) | ((Rhi ^ Y) & ~(X & Y) & M)  with X / Y the read's low / high plane and M its not-N plane
// (read T = X & Y: only the low bit is compared, so reference C and T both match) — three three-input operations and ONE
// popcount per 32 nt instead of four operations and a popcount per 16.
__host__ __device__ inline uint32_t bsx_plane_half(uint32_t w, int which)
{
    uint32_t x = (which ? (w >> 1) : w) & 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    return (x | (x >> 8)) & 0xFFFFu;
}
__host__ __device__ inline uint32_t bsx_plane_word(uint32_t w_first, uint32_t w_second, int which) { return (bsx_plane_half(w_first, which) << 16) | bsx_plane_half(w_second, which); }
// mismatch bits (one per nt) of 32 read nt (planes X, Y, not-N plane M) against 32 reference nt (planes rlo, rhi)
__host__ __device__ inline uint32_t bsx_plane_mismatch(uint32_t rlo, uint32_t rhi, uint32_t X, uint32_t Y, uint32_t M)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // one v_bitop3_b32 per step (truth tables over 0xF0 / 0xCC / 0xAA); left to itself the compiler spends a fourth instruction on X & Y
    const uint32_t a = __builtin_amdgcn_bitop3_b32(rlo, X, M, 0x28), b = __builtin_amdgcn_bitop3_b32(rhi, Y, X, 0x34);
    return __builtin_amdgcn_bitop3_b32(a, b, M, 0xF8);
#else
    const uint32_t a = (rlo ^ X) & M, b = (rhi ^ Y) & ~(X & Y);
    return a | (b & M);
#endif
}

// the same for a word of the read without N and without the read's end (M all ones): two instructions
__host__ __device__ inline uint32_t bsx_plane_mismatch_full(uint32_t rlo, uint32_t rhi, uint32_t X, uint32_t Y)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_bitop3_b32(rhi, Y, X, 0x34), rlo, X, 0xF6);   // b | (rlo ^ X)
#else
    return (rlo ^ X) | ((rhi ^ Y) & ~(X & Y));
#endif
}

// The scan kernel keeps, per task, the read pre-shifted by every s = 0..31 nt (s = candidate position mod 32), so that a
// candidate's reference pairs are compared where they lie — the reference's own scheme (16 shifted copies of the read,
// align.cpp:107-161) on 32-nt planes.  Word j of the read shifted by s: its nt i sits at frame position s + i.
__host__ __device__ inline uint32_t bsx_plane_shift(uint32_t prev, uint32_t cur, uint32_t s) { return s ? (prev << (32 - s)) | (cur >> s) : cur; }
// The reference's CountMismatch works on 64-bit words of its own 16-nt grid: the word that holds the read's first nt starts at
// frame position 0 when s < 16 and at 16 when s >= 16.  Its first early-out looks at frame word 0 plus, for s >= 16, the first
// 16 nt of word 1 (this mask); its second early-out at the same one word further (align.h:189-197).
__host__ __device__ inline uint32_t bsx_plane_bmask(uint32_t s) { return (s & 16u) ? 0xFFFF0000u : 0u; }

// deterministic pick used for equal-best hits: reference utilities.cpp:44-48 (the -S != 0 branch)
__host__ __device__ inline uint32_t bsx_myrand(uint32_t index, int32_t randseed)
{
    uint64_t v = ((uint64_t)(int64_t)(int32_t)index + (uint64_t)(int64_t)(int32_t)(randseed * 1000000)) * 3935559000370003845ull + 2691343689449507681ull;
    v ^= v >> 21; v ^= v << 37; v ^= v >> 4;
    v *= 4768777513237032717ull;
    v ^= v << 20; v ^= v >> 41; v ^= v << 5;
    return (uint32_t)v;
}

// Which block of the scan order a block of the grid takes.  The grid's blocks go to the 8 XCDs round-robin (its size is a multiple
// of 8) and every XCD has its own L2: neighbours in the order should share one.  mode 1 gives every XCD one contiguous eighth of the
// order — but the order is sorted by index entry, the tasks of different buckets differ in cost, and the XCD with the expensive eighth
// finishes long after the others (k_hscan 85 ms per step).  mode N >= 2 deals pieces of N blocks to the XCDs in turn: neighbours still
// share an L2 and every XCD gets a sample of the whole order (N = 128: 69.5 ms; 16 / 32 / 64 / 256: 80.1 / 75.2 / 70.6 / 70.5).
// Returns 0 = take block `b`, 1 = nothing for this grid block at this stride, 2 = the order is exhausted.
__host__ __device__ inline int bsx_order_block(uint32_t vb, uint32_t nvb, uint32_t mode, uint32_t &b)
{
    if (mode == 1) {
        const uint32_t per_xcd = (nvb + 7u) >> 3;
        if ((vb >> 3) >= per_xcd) return 2;
        b = (vb & 7u) * per_xcd + (vb >> 3);
        return b < nvb ? 0 : 1;
    }
    if (mode >= 2) {
        const uint32_t local = vb >> 3, piece = local / mode, within = local - piece * mode;
        if (piece * 8u * mode >= nvb) return 2;
        b = (piece * 8u + (vb & 7u)) * mode + within;
        return b < nvb ? 0 : 1;
    }
    b = vb;
    return vb < nvb ? 0 : 2;
}
