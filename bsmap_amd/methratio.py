#!/usr/bin/env python3
"""methratio — methylation ratios from BSMAP alignments, pile-up on the GPU.

Host-side mirror of the reference's methratio.py (same option letters, same input formats, same table, same summary
line); the per-alignment work — duplicate removal, fill-in trimming, the counter updates under every reference C/G —
and the selection of table rows run in HIP kernels behind the C ABI (include/bsx.h, bsx_meth_*).  There is no CPU
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

_STRAND = {"++": 0, "-+": 1, "+-": 2, "--": 3}


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


class _Batch:
    def __init__(self):
        self.chr, self.pos, self.strand, self.insert, self.cut, self.seqs, self.off = [], [], [], [], [], [], [0]

    def add(self, c, pos, st, ins, cut, seq):
        self.chr.append(c); self.pos.append(pos); self.strand.append(st); self.insert.append(ins); self.cut.append(cut)
        self.seqs.append(seq); self.off.append(self.off[-1] + len(seq))

    def __len__(self):
        return len(self.chr)


def run(reffile, infiles, outfile, chroms=None, unique=False, pair=False, meth0=False, rm_dup=False, trim_fillin=2, combine_CpG=False,
        min_depth=1, device=0, quiet=True, batch=1 << 20):
    """returns the summary line the reference prints on stdout"""
    def disp(txt):
        if not quiet:
            sys.stderr.write("@ %s: %s\n" % (time.asctime(), txt))

    L = _bind()
    disp("reading reference %s ..." % reffile)
    ref = load_reference(reffile, chroms or [])
    names = list(ref.keys())                      # ids follow the FASTA order; the table is written in sorted order
    cid = {n: i for i, n in enumerate(names)}
    lens = np.array([len(ref[n]) for n in names], np.uint64)
    h = C.c_void_p()
    _check(L.bsx_meth_create(len(names), lens.ctypes.data, 1 if rm_dup else 0, device, C.byref(h)))
    try:
        for n in names:
            _check(L.bsx_meth_set_reference(h, cid[n], ref[n].encode("latin-1")))

        def flush(b):
            if not len(b):
                return
            arr = [np.array(b.chr, np.uint32), np.array(b.pos, np.int64), np.array(b.strand, np.uint8), np.array(b.insert, np.int32),
                   np.array(b.cut, np.int64), np.frombuffer("".join(b.seqs).encode("latin-1") + b"\0", np.uint8), np.array(b.off, np.uint64)]
            _check(L.bsx_meth_add(h, len(b), *[a.ctypes.data for a in arr], max(0, trim_fillin)))

        for infile in infiles:
            disp("reading %s ..." % infile)
            ext = infile[-4:].upper()
            if ext == ".BAM":
                raise SystemExit("BAM input is not supported; convert to SAM (the reference pipes it through samtools view)")
            sam = ext == ".SAM"
            b = _Batch()
            with open(infile) as fin:
                for line in fin:  # get_alignment's filters (methratio.py:31-48); the rest of it runs on the device
                    col = line.split("\t")
                    if sam:
                        if line[0] == "@":
                            continue
                        flag = int(col[1])
                        if flag & 0x4 or (unique and flag & 0x100) or (pair and not flag & 0x2):
                            continue
                        cr, pos, seq, strand, insert = col[2], int(col[3]) - 1, col[9], "", int(col[8])
                        if cr not in cid:
                            continue
                        for aux in col[11:]:
                            if aux[:5] == "ZS:Z:":
                                strand = aux[5:7]
                                break
                        if strand == "":
                            raise ValueError("alignment without ZS:Z: tag")
                        cut = int(col[7]) - 1 if insert > 0 else -1
                    else:
                        flag = col[3][:2]
                        if flag == "NM" or flag == "QC" or (unique and flag != "UM") or (pair and col[7] == "0"):
                            continue
                        seq, strand, cr, pos, insert, cut = col[1], col[6], col[4], int(col[5]) - 1, int(col[7]), -1
                        if cr not in cid:
                            continue
                    b.add(cid[cr], pos, _STRAND[strand], insert, cut, seq)
                    if len(b) >= batch:
                        flush(b); b = _Batch()
            flush(b)
        if combine_CpG:
            disp("combining CpG methylation from both strands ...")
            _check(L.bsx_meth_combine_cpg(h))
        disp("writing %s ..." % outfile)
        ss = {"C": "+", "G": "-"}
        z95, z95sq = 1.96, 1.96 * 1.96
        nc = nd = 0
        with open(outfile, "w") as fout:
            fout.write("chr\tpos\tstrand\tcontext\tratio\ttotal_C\tmethy_C\tCI_lower\tCI_upper\n")
            for cr in sorted(names):
                n_rows, cov, sdep = C.c_uint32(), C.c_uint64(), C.c_uint64()
                _check(L.bsx_meth_report_chr(h, cid[cr], min_depth, 1 if meth0 else 0, C.byref(n_rows), C.byref(cov), C.byref(sdep)))
                nc += cov.value; nd += sdep.value
                if not n_rows.value:
                    continue
                pos, dep, met = (np.zeros(n_rows.value, np.uint32) for _ in range(3))
                _check(L.bsx_meth_fetch_rows(h, pos.ctypes.data, dep.ctypes.data, met.ctypes.data))
                refcr = ref[cr]
                out = []
                for i, d, m in zip(pos.tolist(), dep.tolist(), met.tolist()):  # methratio.py:143-151, same arithmetic
                    ratio = float(m) / d
                    pmid = ratio + z95sq / (2 * d)
                    sd = z95 * ((ratio * (1 - ratio) / d + z95sq / (4 * d * d)) ** 0.5)
                    norminator = 1 + z95sq / d
                    out.append("%s\t%d\t%c\t%s\t%.3f\t%d\t%d\t%.3f\t%.3f\n" % (cr, i + 1, ss[refcr[i]], refcr[i - 2:i + 3], ratio, d, m,
                                                                              (pmid - sd) / norminator, (pmid + sd) / norminator))
                fout.write("".join(out))
        nmap = C.c_uint64()
        _check(L.bsx_meth_valid_mappings(h, C.byref(nmap)))
        disp("done.")
        # (with nothing covered the reference dies here on a division by zero; that is reported instead)
        if nc == 0:
            return "total %d valid mappings, 0 covered cytosines.\n" % nmap.value
        return "total %d valid mappings, %d covered cytosines, average coverage: %.2f fold.\n" % (nmap.value, nc, float(nd) / nc)
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
