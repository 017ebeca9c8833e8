// bsx_textout.h - text output through a shared mapping of the output file (host-only, header-only; used by bsmap_main.cpp).
//
// Buffered writes into ONE file hold the inode's lock: 1 or 14 threads calling pwrite at disjoint offsets move the same 5-6 GB/s into a
// file on tmpfs (tools/microbench/shm_write.cpp), which capped the command line at 16-18 M reads/s of SAM text.  Page faults on a shared
// mapping run in parallel: the file is extended to the end of the new range, the range is mapped, and the pieces are copied in by several
// threads that split the BYTES evenly (whatever the pieces are), at whatever page offset the range starts.
#pragma once
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <utility>
#include <vector>

namespace bsx_textout {

// the ranges map_write calls are copying into right now (several may run at once: the lanes' join copies every part on threads of its
// own), for a SIGBUS handler that has to tell a full output file system from a fault on some other mapping (the memory-mapped input files)
constexpr int MAP_SLOTS = 64;
inline std::atomic<uintptr_t> g_map_lo[MAP_SLOTS], g_map_hi[MAP_SLOTS];
inline bool in_mapped_output(uintptr_t a)
{
    for (int i = 0; i < MAP_SLOTS; i++) {
        const uintptr_t lo = g_map_lo[i].load(std::memory_order_relaxed), hi = g_map_hi[i].load(std::memory_order_relaxed);
        if (lo && a >= lo && a < hi) return true;
    }
    return false;
}
// SIGBUS: a full file system under a shared mapping of the output, or an input file truncated while it was read — said, then exit(1)
inline void install_sigbus_handler()
{
    struct sigaction sa; memset(&sa, 0, sizeof(sa));
    sa.sa_flags = SA_SIGINFO;
    sa.sa_sigaction = [](int, siginfo_t *si, void *) {
        static const char m_out[] = "write error on the output file (no space left?)\n", m_other[] = "bus error on a mapped file (an input file truncated while it was read?)\n";
        const bool out = in_mapped_output((uintptr_t)si->si_addr);
        ssize_t r = write(2, out ? m_out : m_other, (out ? sizeof(m_out) : sizeof(m_other)) - 1); (void)r;
        _exit(1);
    };
    sigaction(SIGBUS, &sa, nullptr);
}

// copy the pieces [p_i, p_i + n_i) to consecutive offsets of fd starting at `at`, with up to `nthreads` threads (at least 1 MB each);
// false: nothing was written (the file cannot be extended or mapped) - the caller falls back to pwrite
inline bool map_write(int fd, const std::vector<std::pair<const char *, size_t>> &pieces, off_t at, int nthreads)
{
    size_t total = 0;
    for (auto &x : pieces) total += x.second;
    if (!total) return true;
    {   // extend the file to the end of the new range (never shorten it: ranges of one file may be written concurrently)
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        if (st.st_size < at + (off_t)total && ftruncate(fd, at + (off_t)total) != 0) return false;
    }
    const long pg = sysconf(_SC_PAGESIZE);
    const off_t base = at / pg * pg;
    const size_t lead = (size_t)(at - base), len = lead + total;
    char *m = (char *)mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, base);
    if (m == MAP_FAILED) return false;
    int slot = -1;   // (a free slot of the registry; none free: the copy goes on, a fault would be reported as "a mapped file")
    for (int i = 0; i < MAP_SLOTS && slot < 0; i++) { uintptr_t z = 0; if (g_map_lo[i].compare_exchange_strong(z, (uintptr_t)m)) { g_map_hi[i].store((uintptr_t)m + len, std::memory_order_relaxed); slot = i; } }
    std::vector<size_t> start(pieces.size() + 1, 0);
    for (size_t i = 0; i < pieces.size(); i++) start[i + 1] = start[i] + pieces[i].second;
    auto copy_range = [&](size_t lo, size_t hi) {
        for (size_t i = 0; i < pieces.size() && start[i] < hi; i++) {
            const size_t a = std::max(lo, start[i]), b = std::min(hi, start[i + 1]);
            if (a < b) memcpy(m + lead + a, pieces[i].first + (a - start[i]), b - a);
        }
    };
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, nthreads), total >> 20));
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(copy_range, total * t / T, total * (t + 1) / T);
    copy_range(0, total / T);
    for (std::thread &x : th) x.join();
    if (slot >= 0) { g_map_hi[slot].store(0, std::memory_order_relaxed); g_map_lo[slot].store(0, std::memory_order_relaxed); }
    munmap(m, len);
    return true;
}

}  // namespace bsx_textout
