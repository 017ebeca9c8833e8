#!/usr/bin/env python3
"""methratio — methylation ratios from BSMAP alignments, pile-up on the GPU.

Host-side mirror of the reference's methratio.py (same option letters, same input formats, same table, same summary
line).  This file keeps the option surface; the FASTA and the mapping files are parsed by host threads inside
libbsx.so (bsx_meth_create_from_fasta, bsx_meth_add_file), the per-alignment work — duplicate removal, fill-in trimming, the counter updates under
every reference C/G — and the selection of table rows run in HIP kernels, and the table is formatted by host threads
(bsx_meth_write_table), all behind the C ABI of include/bsx.h.  There is no CPU
fallback: without the library and a gfx950 device this fails.

    python -m bsmap_amd.methratio -o out.txt -d genome.fa [options] alignments.bsp|.sam [...]

Differences from the reference: SAM and BAM files are read directly (numeric flags: 0x4 = 'u', 0x100 = 's', 0x2 = 'P' of
`samtools view -X`; BGZF through zlib), no samtools is spawned and `-s` is accepted and ignored."""
import ctypes as C
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
            nl = C.c_uint64()
            _check(L.bsx_meth_add_file(h, infile.encode(), 2 if ext == ".BAM" else 1 if ext == ".SAM" else 0, None, 1 if unique else 0, 1 if pair else 0, max(0, trim_fillin), C.byref(nl)))
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
    """command line with the reference's option letters (methratio.py:3-17)"""
    import argparse
    ap = argparse.ArgumentParser(prog="methratio", description="methylation ratios from BSMAP mapping files (BSP, SAM or BAM), pile-up on the GPU")
    ap.add_argument("-o", "--out", dest="outfile", required=True, help="table to write")
    ap.add_argument("-d", "--ref", dest="reffile", required=True, help="reference FASTA the reads were mapped to")
    ap.add_argument("-c", "--chr", dest="chroms", default="", help="comma-separated sequence names to keep (default: every sequence)")
    ap.add_argument("-s", "--sam-path", dest="sam_path", default="", help="ignored: SAM files are parsed directly, samtools is not needed")
    ap.add_argument("-u", "--unique", action="store_true", help="use uniquely mapped reads / pairs only")
    ap.add_argument("-p", "--pair", action="store_true", help="use properly paired mappings only")
    ap.add_argument("-z", "--zero-meth", dest="meth0", action="store_true", help="also list covered cytosines without a methylated read")
    ap.add_argument("-q", "--quiet", action="store_true", help="no progress lines on stderr")
    ap.add_argument("-r", "--remove-duplicate", dest="rm_dup", action="store_true", help="keep the first read per fragment end and direction")
    ap.add_argument("-t", "--trim-fillin", dest="trim_fillin", type=int, default=2, help="end-repair nucleotides to ignore at fragment ends (default 2)")
    ap.add_argument("-g", "--combine-CpG", dest="combine_CpG", action="store_true", help="add the counts of the G of each CpG to its C")
    ap.add_argument("-m", "--min-depth", dest="min_depth", type=int, default=1, help="lowest depth a listed cytosine must have (default 1)")
    ap.add_argument("-G", "--gpu", dest="device", type=int, default=0, help="GPU ordinal (extension)")
    ap.add_argument("infiles", nargs="+", help="mapping files written by bsmap: *.sam / *.bam by their format, anything else as BSP")
    o = ap.parse_args(argv)
    sys.stdout.write(run(o.reffile, o.infiles, o.outfile, chroms=o.chroms.split(",") if o.chroms else None, unique=o.unique, pair=o.pair, meth0=o.meth0,
                         rm_dup=o.rm_dup, trim_fillin=o.trim_fillin, combine_CpG=o.combine_CpG, min_depth=o.min_depth, device=o.device, quiet=o.quiet))


if __name__ == "__main__":
    main()
