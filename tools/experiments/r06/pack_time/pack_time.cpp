#include PACK_SRC
#include <chrono>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
int main(int argc, char **argv) {
    bsx_params p; bsx_params_default(&p); bsx_params_finish(&p);
    int fd = open(argv[1], O_RDONLY); struct stat st; fstat(fd, &st);
    const char *m = (const char *)mmap(nullptr, st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    for (int rep = 0; rep < 2; rep++) {
        bsx_ref r; std::vector<uint32_t> a, b;
        auto t0 = std::chrono::steady_clock::now();
        int rc = bsx_pack_fasta(p, m, st.st_size, r, a, b);
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("rc %d  %.3f s  words %zu blocks %zu cpus %u\n", rc, s, a.size(), r.blocks.size(), bsx_usable_cpus());
    }
}
