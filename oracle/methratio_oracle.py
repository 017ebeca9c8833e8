"""CPU restatement of the reference's methylation-ratio pile-up (methratio.py) — TEST INFRASTRUCTURE ONLY: imported by
tests/ (and by nothing under bsmap_amd/).  Pinned against tests/golden/methratio.json.gz, which holds the output of the
reference script itself (converted with lib2to3 at generation time, tests/golden/make_golden_methratio.py) on BSP files
and SAM files written by the real bsmap binary (the script reads SAM through `samtools view -X`: the vendored samtools
0.1.7a front end is built for that by `make -C oracle samtools`).  Parity pinned for both input formats.

Plain Python, small inputs only.  Citations: methratio.py line numbers of the reference tree."""
import numpy as np


class Options:
    def __init__(self, chroms=None, unique=False, pair=False, meth0=False, rm_dup=False, trim_fillin=2, combine_CpG=False, min_depth=1):
        self.chroms, self.unique, self.pair, self.meth0 = chroms, unique, pair, meth0
        self.rm_dup, self.trim_fillin, self.combine_CpG, self.min_depth = rm_dup, trim_fillin, combine_CpG, min_depth


def load_reference(text, chroms=None):
    """methratio.py:67-77: name = first token of the header, sequence upper-cased, optional -c filter"""
    ref, cr, seq = {}, "", []
    for line in text.splitlines(True):
        if line[0] == ">":
            if cr and (not chroms or cr in chroms):
                ref[cr] = "".join(seq).upper()
            cr, seq = line[1:-1].split()[0], []
        else:
            seq.append(line.strip())
    if not chroms or cr in chroms:
        ref[cr] = "".join(seq).upper()
    return ref


def get_alignment(line, o, chroms, coverage, sam_format=False):
    """methratio.py:31-65; returns None or (seq, strand, chr, pos)"""
    col = line.split("\t")
    if sam_format:
        flag = int(col[1])  # the reference reads letter flags from `samtools view -X`: u = 0x4, s = 0x100, P = 0x2
        if flag & 0x4:
            return None
        if o.unique and flag & 0x100:
            return None
        if o.pair and not flag & 0x2:
            return None
        cr, pos, seq, strand, insert = col[2], int(col[3]) - 1, col[9], "", int(col[8])
        if cr not in chroms:
            return None
        for aux in col[11:]:
            if aux[:5] == "ZS:Z:":
                strand = aux[5:7]
                break
        if strand == "":
            raise ValueError
    else:
        flag = col[3][:2]
        if flag == "NM" or flag == "QC":
            return None
        if o.unique and flag != "UM":
            return None
        if o.pair and col[7] == "0":
            return None
        seq, strand, cr, pos, insert = col[1], col[6], col[4], int(col[5]) - 1, int(col[7])
        if cr not in chroms:
            return None
    if o.rm_dup:  # methratio.py:50-54
        if strand == "+-" or strand == "-+":
            frag_end, direction = pos + len(seq), 2
        else:
            frag_end, direction = pos, 1
        if coverage[cr][frag_end] & direction:
            return None
        coverage[cr][frag_end] |= direction
    t = o.trim_fillin
    if t > 0:  # methratio.py:55-63
        if strand == "+-":
            seq = seq[:-t]
        elif strand == "--":
            seq, pos = seq[t:], pos + t
        elif insert != 0 and len(seq) > abs(insert) - t:
            trim_nt = len(seq) - (abs(insert) - t)
            if strand == "++":
                seq = seq[:-trim_nt]
            elif strand == "-+":
                seq, pos = seq[trim_nt:], pos + trim_nt
    if sam_format and insert > 0:
        seq = seq[:int(col[7]) - 1 - pos]
    return seq, strand[0], cr, pos


def run(fasta_text, infiles, o):
    """infiles: list of (name, text).  Returns (table text, summary line or None when the reference would crash)"""
    ref = load_reference(fasta_text, o.chroms)
    chroms = set(ref.keys())
    meth = {c: np.zeros(len(s), np.uint32) for c, s in ref.items()}
    depth = {c: np.zeros(len(s), np.uint32) for c, s in ref.items()}
    coverage = {c: np.zeros(len(s), np.uint8) for c, s in ref.items()} if o.rm_dup else None
    conv = {"+": ("C", "T"), "-": ("G", "A")}
    nmap = 0
    for name, text in infiles:
        sam = name[-4:].upper() == ".SAM"
        for line in text.splitlines(True):
            if sam and line.startswith("@"):
                continue
            a = get_alignment(line, o, chroms, coverage, sam)
            if a is None:
                continue
            seq, strand, cr, pos = a
            if pos + len(seq) > len(depth[cr]):
                continue
            nmap += 1
            refseq = ref[cr][pos:pos + len(seq)]
            match, convert = conv[strand]
            index = refseq.find(match)
            while index >= 0:  # methratio.py:108-114
                if seq[index] == convert:
                    depth[cr][pos + index] += 1
                elif seq[index] == match:
                    meth[cr][pos + index] += 1
                    depth[cr][pos + index] += 1
                index = refseq.find(match, index + 1)
    if o.combine_CpG:  # methratio.py:118-128
        for cr in depth:
            pos = ref[cr].find("CG")
            while pos >= 0:
                depth[cr][pos] += depth[cr][pos + 1]
                meth[cr][pos] += meth[cr][pos + 1]
                depth[cr][pos + 1] = 0
                meth[cr][pos + 1] = 0
                pos = ref[cr].find("CG", pos + 2)
    ss = {"C": "+", "G": "-"}
    z95, z95sq = 1.96, 1.96 * 1.96
    out = ["chr\tpos\tstrand\tcontext\tratio\ttotal_C\tmethy_C\tCI_lower\tCI_upper\n"]
    nc = nd = 0
    for cr in sorted(depth.keys()):  # methratio.py:135-151
        refcr = ref[cr]
        for i in np.nonzero(depth[cr] >= max(1, o.min_depth))[0] if o.min_depth >= 1 else range(len(refcr)):
            i = int(i)
            d = int(depth[cr][i])
            if d < o.min_depth:
                continue
            nc += 1
            nd += d
            m = int(meth[cr][i])
            if m == 0 and not o.meth0:
                continue
            ratio = float(m) / d
            seq = refcr[i - 2:i + 3]
            pmid = ratio + z95sq / (2 * d)
            sd = z95 * ((ratio * (1 - ratio) / d + z95sq / (4 * d * d)) ** 0.5)
            norminator = 1 + z95sq / d
            CIl, CIu = (pmid - sd) / norminator, (pmid + sd) / norminator
            out.append("%s\t%d\t%c\t%s\t%.3f\t%d\t%d\t%.3f\t%.3f\n" % (cr, i + 1, ss[refcr[i]], seq, ratio, d, m, CIl, CIu))
    summary = None if nc == 0 else "total %d valid mappings, %d covered cytosines, average coverage: %.2f fold.\n" % (nmap, nc, float(nd) / nc)
    return "".join(out), summary


def options_from_argv(argv):
    """the reference's option letters (methratio.py:3-17) -> Options"""
    o = Options()
    i = 0
    while i < len(argv):
        a = argv[i]
        if a == "-u": o.unique = True
        elif a == "-p": o.pair = True
        elif a == "-z": o.meth0 = True
        elif a == "-r": o.rm_dup = True
        elif a == "-g": o.combine_CpG = True
        elif a == "-t": i += 1; o.trim_fillin = int(argv[i])
        elif a == "-m": i += 1; o.min_depth = int(argv[i])
        elif a == "-c": i += 1; o.chroms = argv[i].split(",")
        else: raise ValueError(a)
        i += 1
    return o
