// Host microbenchmark: how fast can N threads put text into ONE file on /dev/shm?  (a) pwrite at disjoint offsets (what the command line's
// write stage does), (b) memcpy into a MAP_SHARED mapping of the file, (c) pwrite into N separate files (no shared inode), (d) as (b) with the piece's pages
// populated by one madvise(MADV_POPULATE_WRITE) before the copy instead of a write fault per page, for N = 1..16.
// usage: shm_write [GiB=4] [dir=/dev/shm]     (output: one JSON object)
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const size_t total = (size_t)(argc > 1 ? atof(argv[1]) : 4.0) << 30;
    const std::string dir = argc > 2 ? argv[2] : "/dev/shm";
    const size_t piece = 64u << 20;  // a worker's chunk of a batch is about this size
    std::vector<char> src(piece);
    for (size_t i = 0; i < piece; i++) src[i] = (char)('A' + i % 23);
    printf("{\"bytes\": %zu, \"piece\": %zu, \"runs\": [", total, piece);
    bool first = true;
    for (int mode = 0; mode < 4; mode++)
        for (int n : {1, 2, 4, 8, 14, 16}) {
            const std::string path = dir + "/bsx_shm_write_test";
            std::vector<int> fds;
            const int nf = mode == 2 ? n : 1;
            for (int f = 0; f < nf; f++) {
                const int fd = open((path + std::to_string(f)).c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
                if (fd < 0) { perror("open"); return 1; }
                fds.push_back(fd);
            }
            char *map = nullptr;
            if (mode == 1 || mode == 3) {
                if (ftruncate(fds[0], (off_t)total)) { perror("ftruncate"); return 1; }
                map = (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fds[0], 0);
                if (map == MAP_FAILED) { perror("mmap"); return 1; }
            }
            const size_t n_pieces = total / piece;
            const double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < n; t++)
                th.emplace_back([&, t] {
                    for (size_t p = t; p < n_pieces; p += n) {
                        if (mode == 1) memcpy(map + p * piece, src.data(), piece);
                        else if (mode == 3) { if (madvise(map + p * piece, piece, 23 /* MADV_POPULATE_WRITE */) != 0) { perror("madvise"); exit(1); } memcpy(map + p * piece, src.data(), piece); }
                        else {
                            const int fd = mode == 2 ? fds[t] : fds[0];
                            const off_t at = mode == 2 ? (off_t)((p / n) * piece) : (off_t)(p * piece);
                            size_t done = 0;
                            while (done < piece) { const ssize_t w = pwrite(fd, src.data() + done, piece - done, at + (off_t)done); if (w <= 0) { perror("pwrite"); exit(1); } done += (size_t)w; }
                        }
                    }
                });
            for (auto &x : th) x.join();
            const double dt = now() - t0;
            if (map) munmap(map, total);
            for (int f = 0; f < nf; f++) { close(fds[f]); unlink((path + std::to_string(f)).c_str()); }
            printf("%s{\"mode\": \"%s\", \"threads\": %d, \"GBps\": %.2f}", first ? "" : ", ", mode == 0 ? "pwrite, one file" : mode == 1 ? "memcpy into a shared mapping" : mode == 2 ? "pwrite, one file per thread" : "populate + memcpy into a shared mapping", n, total / dt / 1e9);
            first = false;
            fflush(stdout);
        }
    printf("]}\n");
    return 0;
}
