// bsx_synth.hip — device-side synthetic workload generators (bench input).  Filled in below.
#include "bsx_internal.h"
extern "C" int bsx_ref_create_synthetic(const bsx_params *p, uint32_t n_chr, const uint32_t *chr_len, uint64_t seed, int device, bsx_ref **out)
{ (void)p; (void)n_chr; (void)chr_len; (void)seed; (void)device; (void)out; return BSX_ERR_STATE; }
extern "C" int bsx_batch_synth_reads(bsx_batch *b, uint32_t n, uint32_t read_len, uint64_t seed, uint32_t first_index)
{ (void)b; (void)n; (void)read_len; (void)seed; (void)first_index; return BSX_ERR_STATE; }
