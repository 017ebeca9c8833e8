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

// deterministic pick used for equal-best hits: reference utilities.cpp:44-48 (the -S != 0 branch)
__host__ __device__ inline uint32_t bsx_myrand(uint32_t index, int32_t randseed)
{
    uint64_t v = ((uint64_t)(int64_t)(int32_t)index + (uint64_t)(int64_t)(int32_t)(randseed * 1000000)) * 3935559000370003845ull + 2691343689449507681ull;
    v ^= v >> 21; v ^= v << 37; v ^= v >> 4;
    v *= 4768777513237032717ull;
    v ^= v << 20; v ^= v >> 41; v ^= v << 5;
    return (uint32_t)v;
}
