// Host-side check of the bit-plane comparison the scan kernels use (bsx_dev.h) against the per-nt definition of the
// reference's mismatch rule (align.h:167-200, param.h:125-147): for random reads (with N and T), random references and every
// candidate position, the plane path — reference planes, the read pre-shifted by p mod 32, frame words 0..5, the B mask —
// must give the same total, the same first-early-out count (nt [0, 32 - p % 16)) and the same second one (nt [0, 64 - p % 16)).
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include "../../bsmap_amd/csrc/bsx_dev.h"

static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 11); }

int main()
{
    const int REF_NT = 4096;
    long checked = 0;
    for (int trial = 0; trial < 400; trial++) {
        std::vector<uint8_t> ref(REF_NT + 512);
        for (auto &c : ref) c = rnd() & 3;
        // packed words: 16 nt per word, first nt in the top bits
        std::vector<uint32_t> words((ref.size() + 15) / 16, 0);
        for (size_t i = 0; i < ref.size(); i++) words[i >> 4] |= (uint32_t)ref[i] << (30 - 2 * (i & 15));
        std::vector<uint32_t> planes(2 * ((words.size() + 1) / 2) + 32, 0);
        for (size_t g = 0; 2 * g + 1 < words.size(); g++) { planes[2 * g] = bsx_plane_word(words[2 * g], words[2 * g + 1], 0); planes[2 * g + 1] = bsx_plane_word(words[2 * g], words[2 * g + 1], 1); }
        const int len = 16 + (int)(rnd() % 129);  // 16..144
        std::vector<int> rd(len);
        for (auto &c : rd) { const uint32_t r = rnd() % 20; c = r == 0 ? -1 : (int)(r & 3); }  // -1 = N
        // the read as the kernels hold it: 10 packed words + masks, then planes (publish_window)
        uint32_t w[10] = {0}, m[10] = {0};
        for (int i = 0; i < len; i++) { w[i >> 4] |= (uint32_t)(rd[i] < 0 ? 0 : rd[i]) << (30 - 2 * (i & 15)); m[i >> 4] |= (rd[i] < 0 ? 0u : 3u) << (30 - 2 * (i & 15)); }
        uint32_t px[7] = {0}, py[7] = {0}, pm[7] = {0};  // [j + 1] = plane word j
        for (int j = 0; j < 5; j++) { px[j + 1] = bsx_plane_word(w[2 * j], w[2 * j + 1], 0); py[j + 1] = bsx_plane_word(w[2 * j], w[2 * j + 1], 1); pm[j + 1] = bsx_plane_word(m[2 * j], m[2 * j + 1], 0); }
        const int thres_words = (len + 31 + 31) >> 5;  // nW of the kernel
        for (int p = 16; p < REF_NT; p += 1 + (int)(rnd() % 3)) {
            // per-nt definition
            int tot = 0, w0 = 0, w01 = 0;
            const int k = p & 15;
            for (int i = 0; i < len; i++) {
                const int r = rd[i], c = ref[p + i];
                const bool mis = r >= 0 && !(r == c || (r == 3 && c == 1));
                tot += mis; if (i < 32 - k) w0 += mis; if (i < 64 - k) w01 += mis;
            }
            // plane path
            const uint32_t s = (uint32_t)p & 31u, g = (uint32_t)p >> 5, B = bsx_plane_bmask(s);
            uint32_t mm[6];
            for (int j = 0; j < 6; j++) {
                const uint32_t X = bsx_plane_shift(px[j], px[j + 1], s), Y = bsx_plane_shift(py[j], py[j + 1], s), M = bsx_plane_shift(pm[j], pm[j + 1], s);
                if (j >= thres_words && M) { printf("FAIL nW: len %d s %u word %d has mask bits\n", len, s, j); return 1; }
                mm[j] = bsx_plane_mismatch(planes[2 * (g + j)], planes[2 * (g + j) + 1], X, Y, M);
            }
            const int c0 = __builtin_popcount(mm[0]);
            const int w0p = c0 + __builtin_popcount(mm[1] & B), p64 = c0 + __builtin_popcount(mm[1]);
            const int w01p = p64 + __builtin_popcount(mm[2] & B);
            int totp = p64;
            for (int j = 2; j < 6; j++) totp += __builtin_popcount(mm[j]);
            if (totp != tot || w0p != w0 || w01p != w01 || !(w0p <= p64 && p64 <= w01p)) {
                printf("FAIL trial %d len %d p %d: total %d/%d w0 %d/%d w01 %d/%d p64 %d\n", trial, len, p, totp, tot, w0p, w0, w01p, w01, p64);
                return 1;
            }
            // the read-frame form of k_hscan_shared: the candidate's pairs funnel-shifted to the read (one pair early when aligned)
            {
                const uint32_t pm1 = (uint32_t)p - 1u, kp = pm1 >> 5, shf = 31u - (pm1 & 31u), him = 0xFFFFFFFFu << ((pm1 + 1u) & 15u);
                auto alignbit = [](uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31u)); };
                const int nwr = (len + 31) >> 5;
                int totr = 0, w0r = 0, w01r = 0;
                for (int t = 0; t < nwr; t++) {
                    const uint32_t flo = alignbit(planes[2 * (kp + t)], planes[2 * (kp + t + 1)], shf), fhi = alignbit(planes[2 * (kp + t) + 1], planes[2 * (kp + t + 1) + 1], shf);
                    const uint32_t mmr = bsx_plane_mismatch(flo, fhi, px[t + 1], py[t + 1], pm[t + 1]);
                    totr += __builtin_popcount(mmr);
                    if (t == 0) { w0r = __builtin_popcount(mmr & him); w01r = __builtin_popcount(mmr); }
                    if (t == 1) w01r += __builtin_popcount(mmr & him);
                }
                if (totr != tot || w0r != w0 || w01r != w01) { printf("FAIL read frame: len %d p %d: total %d/%d w0 %d/%d w01 %d/%d\n", len, p, totr, tot, w0r, w0, w01r, w01); return 1; }
            }
            checked++;
        }
    }
    printf("ok %ld candidates\n", checked);
    return 0;
}
