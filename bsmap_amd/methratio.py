#!/usr/bin/env python3
"""methratio — methylation ratios from BSMAP alignments, pile-up on the GPU.

Host-side mirror of the reference's methratio.py (same option letters, same input formats, same table, same summary
line).  This file keeps the option surface; the FASTA and the mapping files are parsed by host threads inside
libbsx.so (bsx_meth_create_from_fasta, bsx_meth_add_file), the per-alignment work — duplicate removal, fill-in trimming, the counter updates under
every reference C/G — and the selection of table rows run in HIP kernels, and the table is formatted by host threads
(bsx_meth_write_table), all behind the C ABI of include/bsx.h.  There is no CPU
fallback: without the library and a gfx950 device this fails.

    python -m bsmap_amd.methratio -o out.txt -d genome.fa [options] alignments.bsp|.sam [...]

Differences from the reference: SAM files are read directly (numeric flags: 0x4 = 'u', 0x100 = 's', 0x2 = 'P' of
`samtools view -X`), no samtools is spawned and `-s` is accepted and ignored; BAM input is not supported."""
import ctypes as C
import optparse
import sys
import time

import numpy as np

from . import lib, _check

def _bind():
    L = lib()
    if getattr(L, "_meth_bound", False):
        return L
    vp, u32, i32, u64 = C.c_void_p, C.c_uint32, C.c_int32, C.c_uint64
    L.bsx_meth_create.argtypes = [u32, vp, i32, i32, C.POINTER(vp)]
    L.bsx_meth_destroy.argtypes = [vp]; L.bsx_meth_destroy.restype = None
    L.bsx_meth_set_reference.argtypes = [vp, u32, C.c_char_p]
    L.bsx_meth_add.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp, vp, u32]
    L.bsx_meth_combine_cpg.argtypes = [vp]
    L.bsx_meth_valid_mappings.argtypes = [vp, vp]
    L.bsx_meth_report_chr.argtypes = [vp, u32, u32, i32, vp, vp, vp]
    L.bsx_meth_fetch_rows.argtypes = [vp, vp, vp, vp]
    L.bsx_meth_create_from_fasta.argtypes = [C.c_char_p, C.c_char_p, i32, i32, C.POINTER(vp)]
    L.bsx_meth_add_file.argtypes = [vp, C.c_char_p, i32, vp, i32, i32, u32, vp]
    L.bsx_meth_write_table.argtypes = [vp, C.c_char_p, u32, vp, vp, u32, i32, vp, vp]
    L._meth_bound = True
    return L


def load_reference(path, chroms):
    """methratio.py:67-77"""
    ref, cr, seq = {}, "", []
    with open(path) as f:
        for line in f:
            if line[0] == ">":
                if cr and (not chroms or cr in chroms):
                    ref[cr] = "".join(seq).upper()
                cr, seq = line[1:-1].split()[0], []
            else:
                seq.append(line.strip())
    if not chroms or cr in chroms:
        ref[cr] = "".join(seq).upper()
    return ref


def run(reffile, infiles, outfile, chroms=None, unique=False, pair=False, meth0=False, rm_dup=False, trim_fillin=2, combine_CpG=False,
        min_depth=1, device=0, quiet=True):
    """returns the summary line the reference prints on stdout"""
    def disp(txt):
        if not quiet:
            sys.stderr.write("@ %s: %s\n" % (time.asctime(), txt))

    L = _bind()
    disp("reading reference %s ..." % reffile)
    h = C.c_void_p()
    _check(L.bsx_meth_create_from_fasta(reffile.encode(), ",".join(chroms).encode() if chroms else None, 1 if rm_dup else 0, device, C.byref(h)))
    try:
        for infile in infiles:
            disp("reading %s ..." % infile)
            ext = infile[-4:].upper()
            if ext == ".BAM":
                raise SystemExit("BAM input is not supported; convert to SAM (the reference pipes it through samtools view)")
            nl = C.c_uint64()
            _check(L.bsx_meth_add_file(h, infile.encode(), 1 if ext == ".SAM" else 0, None, 1 if unique else 0, 1 if pair else 0, max(0, trim_fillin), C.byref(nl)))
        if combine_CpG:
            disp("combining CpG methylation from both strands ...")
            _check(L.bsx_meth_combine_cpg(h))
        disp("writing %s ..." % outfile)
        nc, nd, nmap = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(L.bsx_meth_write_table(h, outfile.encode(), 0, None, None, min_depth, 1 if meth0 else 0, C.byref(nc), C.byref(nd)))
        _check(L.bsx_meth_valid_mappings(h, C.byref(nmap)))
        disp("done.")
        # (with nothing covered the reference dies here on a division by zero; that is reported instead)
        if nc.value == 0:
            return "total %d valid mappings, 0 covered cytosines.\n" % nmap.value
        return "total %d valid mappings, %d covered cytosines, average coverage: %.2f fold.\n" % (nmap.value, nc.value, float(nd.value) / nc.value)
    finally:
        L.bsx_meth_destroy(h)


def main(argv=None):
    parser = optparse.OptionParser(usage="usage: %prog [options] BSMAP_MAPPING_FILES")
    parser.add_option("-o", "--out", dest="outfile", metavar="FILE", help="output file name. (required)", default="")
    parser.add_option("-d", "--ref", dest="reffile", metavar="FILE", help="reference genome fasta file. (required)", default="")
    parser.add_option("-c", "--chr", dest="chroms", metavar="CHR", help="process only specified chromosomes, separated by ','. [default: all]", default="")
    parser.add_option("-s", "--sam-path", dest="sam_path", metavar="PATH", help="accepted for compatibility (SAM files are read directly)", default="")
    parser.add_option("-u", "--unique", action="store_true", dest="unique", help="process only unique mappings/pairs.", default=False)
    parser.add_option("-p", "--pair", action="store_true", dest="pair", help="process only properly paired mappings.", default=False)
    parser.add_option("-z", "--zero-meth", action="store_true", dest="meth0", help="report loci with zero methylation ratios.", default=False)
    parser.add_option("-q", "--quiet", action="store_true", dest="quiet", help="don't print progress on stderr.", default=False)
    parser.add_option("-r", "--remove-duplicate", action="store_true", dest="rm_dup", help="remove duplicated reads.", default=False)
    parser.add_option("-t", "--trim-fillin", dest="trim_fillin", type="int", metavar="N", help="trim N end-repairing fill-in nucleotides. [default: 2]", default=2)
    parser.add_option("-g", "--combine-CpG", action="store_true", dest="combine_CpG", help="combine CpG methylaion ratios on both strands.", default=False)
    parser.add_option("-m", "--min-depth", dest="min_depth", type="int", metavar="FOLD", help="report loci with sequencing depth>=FOLD. [default: 1]", default=1)
    parser.add_option("-G", "--gpu", dest="device", type="int", metavar="N", help="GPU ordinal (extension). [default: 0]", default=0)
    o, infiles = parser.parse_args(argv)
    if len(o.reffile) == 0: parser.error("Missing reference file, use -d or --ref option.")
    if len(o.outfile) == 0: parser.error("Missing output file name, use -o or --out option.")
    if len(infiles) == 0: parser.error("Require at least one BSMAP_MAPPING_FILE.")
    sys.stdout.write(run(o.reffile, infiles, o.outfile, chroms=o.chroms.split(",") if o.chroms else None, unique=o.unique, pair=o.pair, meth0=o.meth0,
                         rm_dup=o.rm_dup, trim_fillin=o.trim_fillin, combine_CpG=o.combine_CpG, min_depth=o.min_depth, device=o.device, quiet=o.quiet))


if __name__ == "__main__":
    main()
