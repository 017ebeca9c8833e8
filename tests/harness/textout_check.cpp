// test harness for bsmap_amd/csrc/bsx_textout.h: writes a header with pwrite, then several "batches" of pseudo-random pieces through
// map_write at running (page-unaligned) offsets, reads the file back and compares it byte for byte with what a plain sequential writer
// would have produced (exit code 0 and the size on stdout if equal).   usage: textout_check <out file> <seed> <threads> <batches>
#include <fcntl.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include "../../bsmap_amd/csrc/bsx_textout.h"

int main(int argc, char **argv)
{
    if (argc < 5) return 2;
    uint64_t x = strtoull(argv[2], nullptr, 10) * 2654435761u + 12345;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    const int threads = atoi(argv[3]), batches = atoi(argv[4]);
    const int fd = open(argv[1], O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return 3;
    std::string expect = "@HD\tVN:1.0\n@SQ\tSN:chr1\tLN:12345\n";
    if (pwrite(fd, expect.data(), expect.size(), 0) != (ssize_t)expect.size()) return 4;
    off_t at = (off_t)expect.size();
    for (int b = 0; b < batches; b++) {
        const int np = 1 + (int)(rnd() % 17);
        std::vector<std::string> text(np);
        std::vector<std::pair<const char *, size_t>> pieces;
        for (int i = 0; i < np; i++) {
            const size_t n = (rnd() % 5 == 0) ? 0 : (size_t)(rnd() % (b % 3 == 2 ? 3000000 : 70000));  // empty pieces, small and multi-megabyte batches
            text[i].resize(n);
            for (size_t k = 0; k < n; k++) text[i][k] = (char)('!' + (rnd() >> 33) % 90);
            pieces.emplace_back(text[i].data(), text[i].size());
            expect += text[i];
        }
        if (!bsx_textout::map_write(fd, pieces, at, threads)) return 5;
        size_t tot = 0; for (auto &p : pieces) tot += p.second;
        at += (off_t)tot;
    }
    close(fd);
    // read the file back with plain stdio and compare with what a sequential writer would have produced
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 6;
    std::string got(expect.size() + 16, '\0');
    const size_t n = fread(&got[0], 1, got.size(), f);
    fclose(f);
    if (n != expect.size()) { fprintf(stderr, "size %zu, expected %zu\n", n, expect.size()); return 7; }
    got.resize(n);
    if (got != expect) {
        size_t k = 0; while (got[k] == expect[k]) k++;
        fprintf(stderr, "first difference at byte %zu of %zu\n", k, n);
        return 8;
    }
    printf("%zu\n", expect.size());
    return 0;
}
