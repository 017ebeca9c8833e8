// CPU harness for the reference packer of libbsx (csrc/bsx_host.cpp: bsx_pack_fasta), built with tiny chunk sizes so that a small FASTA crosses hundreds of
// chunk borders of the parallel packer.  usage: pack_check <fasta> <out.bin> [digest site, e.g. C-CGG]
// out.bin (little-endian u32 unless noted): n_chr, n_words(u64), n_blocks, then anchor[n_chr+1], chr_size[n_chr], rc_offset[n_chr], refcat[n_words],
// crefcat[n_words], blocks[n_blocks][3], and for RRBS per chromosome: n_sites, sites[n_sites]
#define BSX_PACK_CHUNK 997u
#define BSX_PACK_WORDS 61u
#define BSX_PACK_NXCHUNK 1013u
#define BSX_PACK_BIG 4096u
#define BSX_PACK_MINTEXT 1u
#define BSX_PACK_FEW(ncpu) 1000000u
#include "../../bsmap_amd/csrc/bsx_host.cpp"
#include <cstdio>
int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    bsx_params p;
    bsx_params_default(&p);
    if (argc > 3 && bsx_params_set_digest(&p, argv[3]) != BSX_OK) return 3;
    if (bsx_params_finish(&p) != BSX_OK) return 4;
    std::ifstream f(argv[1], std::ios::binary);
    std::stringstream ss; ss << f.rdbuf();
    const std::string text = ss.str();
    bsx_ref r;
    std::vector<uint32_t> refcat, crefcat;
    const int rc = bsx_pack_fasta(p, text.data(), text.size(), r, refcat, crefcat);
    if (rc != BSX_OK) { fprintf(stderr, "bsx_pack_fasta: %d\n", rc); return 5; }
    FILE *o = fopen(argv[2], "wb");
    if (!o) return 6;
    const uint32_t nc = r.n_chr, nb = (uint32_t)r.blocks.size();
    const uint64_t nw = r.n_words;
    fwrite(&nc, 4, 1, o); fwrite(&nw, 8, 1, o); fwrite(&nb, 4, 1, o);
    fwrite(r.anchor.data(), 4, nc + 1, o); fwrite(r.chr_size.data(), 4, nc, o); fwrite(r.rc_offset.data(), 4, nc, o);
    fwrite(refcat.data(), 4, nw, o); fwrite(crefcat.data(), 4, nw, o);
    for (const Block &b : r.blocks) { const uint32_t v[3] = {b.id, b.begin, b.end}; fwrite(v, 4, 3, o); }
    if (p.rrbs) for (uint32_t c = 0; c < nc; c++) { const uint32_t ns = (uint32_t)r.sites[c].size(); fwrite(&ns, 4, 1, o); fwrite(r.sites[c].data(), 4, ns, o); }
    fclose(o);
    return 0;
}
