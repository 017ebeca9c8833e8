// bsx_lanes.h — cutting the read files of a run into lanes (host-only, header-only; used by bsmap_main.cpp and its CPU test harness).
//
// The reference's way to spread one input over several machines is `-B <first read> -E <last read>` (README.txt:83-86): every
// shard opens the same files, skips (first - 1) * 4 lines with getline (reads.cpp:50-76) and maps its range.  A lane here is such
// a shard — one process with its own GPU, reader, formatters and output file — except that the lines are not skipped one by one:
// the parent counts the newlines of the files once, in parallel, and hands every lane the BYTE offset its range starts at.
// The cut is by records of 4 lines (FASTQ) or 2 (FASTA), the same rule as the reference's skip, and every cut is checked to sit
// on a record start ('@' with a '+' line two lines down, or '>'); files that do not keep that layout are not cut (one lane).
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace bsx_lanes {

struct LineIndex {
    const char *base = nullptr;
    size_t len = 0;
    size_t chunk = 0;                 // bytes per counted chunk
    std::vector<uint64_t> before;     // newlines before each chunk (prefix sums), one extra entry = all of them
    uint64_t lines = 0;               // lines of the file: newlines, plus one for an unterminated last line
    int format = -1;                  // 0 FASTQ ('@'), 1 FASTA ('>'), -1 anything else (BAM, empty)
    bool ok() const { return base != nullptr && format >= 0; }
    ~LineIndex() { if (base) munmap(const_cast<char *>(base), len); }
    LineIndex() {}
    LineIndex(const LineIndex &) = delete;
    LineIndex &operator=(const LineIndex &) = delete;
};

inline uint64_t count_nl(const char *p, size_t n)
{
    uint64_t c = 0;
    const char *e = p + n;
    while (p < e) { const char *q = (const char *)memchr(p, '\n', (size_t)(e - p)); if (!q) break; c++; p = q + 1; }
    return c;
}

// newline counts of `path` by chunks, with up to `threads` threads
inline bool index_lines(const std::string &path, int threads, LineIndex &ix)
{
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size <= 0) { ::close(fd); return false; }
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) return false;
    ix.base = (const char *)m; ix.len = (size_t)st.st_size;
    size_t i = 0;
    while (i < ix.len && (ix.base[i] == ' ' || ix.base[i] == '\n' || ix.base[i] == '\t' || ix.base[i] == '\r')) i++;
    ix.format = i < ix.len ? (ix.base[i] == '@' ? 0 : ix.base[i] == '>' ? 1 : -1) : -1;
    if (i != 0) ix.format = -1;   // leading blanks: the line rule and the token rule of the reader would disagree
    ix.chunk = (size_t)4 << 20;
    const size_t nc = (ix.len + ix.chunk - 1) / ix.chunk;
    std::vector<uint64_t> cnt(nc, 0);
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, threads), nc));
    std::vector<std::thread> th;
    auto work = [&](int t) { for (size_t c = (size_t)t; c < nc; c += (size_t)T) cnt[c] = count_nl(ix.base + c * ix.chunk, std::min(ix.chunk, ix.len - c * ix.chunk)); };
    for (int t = 1; t < T; t++) th.emplace_back(work, t);
    work(0);
    for (std::thread &x : th) x.join();
    ix.before.assign(nc + 1, 0);
    for (size_t c = 0; c < nc; c++) ix.before[c + 1] = ix.before[c] + cnt[c];
    ix.lines = ix.before[nc] + (ix.base[ix.len - 1] != '\n' ? 1 : 0);
    return true;
}

// byte offset at which line `line` (0-based) starts; len if the file has no such line
inline size_t line_offset(const LineIndex &ix, uint64_t line)
{
    if (line == 0) return 0;
    if (line > ix.before.back()) return ix.len;   // the start of line k is behind the k-th newline
    size_t c = (size_t)(std::upper_bound(ix.before.begin(), ix.before.end(), line - 1) - ix.before.begin()) - 1;   // chunk that holds newline number `line` (1-based)
    uint64_t need = line - ix.before[c];
    const char *p = ix.base + c * ix.chunk, *e = ix.base + std::min(ix.len, (c + 1) * ix.chunk);
    while (need) { p = (const char *)memchr(p, '\n', (size_t)(e - p)); if (!p) return ix.len; p++; need--; }
    return (size_t)(p - ix.base);
}

// does a record start at byte `off`?  FASTQ: '@' here and '+' at the start of the line two lines down; FASTA: '>'
inline bool record_starts_at(const LineIndex &ix, size_t off)
{
    if (off >= ix.len) return off == ix.len;
    if (ix.format == 1) return ix.base[off] == '>';
    if (ix.base[off] != '@') return false;
    const char *p = ix.base + off, *e = ix.base + ix.len;
    for (int k = 0; k < 2; k++) { p = (const char *)memchr(p, '\n', (size_t)(e - p)); if (!p) return false; p++; }
    return p < e && *p == '+';
}

struct Lane { uint64_t first = 0, count = 0; size_t off_a = 0, off_b = 0; };   // records [first, first + count) (0-based ordinals of the file), byte offsets of record `first`
struct Plan { std::vector<Lane> lanes; uint64_t total = 0; bool mates_differ = false; uint64_t n_a = 0, n_b = 0; std::string why_not; };

// Cut records [read_start - 1, read_end) of the file(s) into `n_lanes` ranges of (nearly) equal size.  `b` may be null (single reads).
// Mate files of unequal length: like the reference (main.cpp:88-93: 50 000 pairs per batch, stop at the first batch whose two counts
// differ) the first floor(min / 50000) * 50000 pairs are mapped.  An empty plan (why_not set) = do not cut.
inline Plan plan_lanes(const LineIndex &a, const LineIndex *b, int n_lanes, uint64_t read_start, uint64_t read_end)
{
    Plan P;
    if (!a.ok() || (b && (!b->ok() || b->format != a.format))) { P.why_not = "input is not FASTA / FASTQ text of one kind"; return P; }
    const uint64_t per = a.format == 0 ? 4 : 2;
    P.n_a = a.lines / per; P.n_b = b ? b->lines / per : P.n_a;
    const uint64_t s0 = read_start - 1;
    uint64_t ea = std::min<uint64_t>(P.n_a, read_end), eb = std::min<uint64_t>(P.n_b, read_end);
    if (ea <= s0 || eb <= s0) { P.why_not = "no read in the requested range"; return P; }
    uint64_t total = std::min(ea, eb) - s0;
    if (ea != eb) { P.mates_differ = true; total = total / 50000 * 50000; if (!total) { P.why_not = "mate files differ in length before the first 50000 pairs"; return P; } }
    // a ragged tail (lines beyond the last whole record) stays with the last lane's reader, which treats it like the single pipeline
    n_lanes = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)n_lanes, total));
    for (int l = 0; l < n_lanes; l++) {
        Lane L;
        L.first = s0 + total * (uint64_t)l / (uint64_t)n_lanes;
        L.count = s0 + total * (uint64_t)(l + 1) / (uint64_t)n_lanes - L.first;
        L.off_a = line_offset(a, L.first * per);
        L.off_b = b ? line_offset(*b, L.first * per) : 0;
        if (!record_starts_at(a, L.off_a) || (b && !record_starts_at(*b, L.off_b))) { P.lanes.clear(); P.why_not = "a cut does not fall on a record start (records are not 4 / 2 lines each)"; return P; }
        P.lanes.push_back(L);
    }
    P.total = total;
    return P;
}

// BSX_P1_EXACT across lanes (bsmap_main.cpp: fork_lanes).  effect[l] = what the reads of lane l do to the reference's never-reset planner state, as 32-bit words with
// 0xFFFFFFFF = "this range leaves the word alone" (the lane's pre-pass run from a state of marker words); have[l] = lane l delivered one.  Returns the state at the
// FIRST read of every lane: word by word the effect of the nearest earlier lane that wrote the word, zero — a fresh aligner object — if none did.
inline std::vector<std::vector<uint32_t>> compose_lane_states(const std::vector<std::vector<uint32_t>> &effect, const std::vector<char> &have, size_t words)
{
    std::vector<std::vector<uint32_t>> start(effect.size(), std::vector<uint32_t>(words, 0u));
    std::vector<uint32_t> st(words, 0u);
    for (size_t l = 0; l < effect.size(); l++) {
        start[l] = st;
        if (have[l]) for (size_t k = 0; k < words; k++) if (effect[l][k] != 0xFFFFFFFFu) st[k] = effect[l][k];
    }
    return start;
}

}  // namespace bsx_lanes
