"""Seeded synthetic genomes and bisulfite reads for the parity tests (small sizes, pure numpy)."""
import numpy as np

COMP = bytes.maketrans(b"ACGTacgtNn", b"TGCAtgcaNn")


def revcomp(s: str) -> str:
    return s.encode().translate(COMP)[::-1].decode()


def random_seq(rng, n, gc=0.5):
    p = [(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2]
    return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.choice(4, size=n, p=p)]


def make_genome(seed=1, chr_lens=(1_000_000,), gc=0.51, n_runs=6, repeats=20, microsats=10, lower=4, iupac=5, cpg_sites=0,
                digest="CCGG"):
    """returns list of (name, str).  Features that the reference treats specially are all present:
    N runs (block breaks), short islands (<30 nt, dropped), lower-case stretches, IUPAC codes (packed as A),
    dispersed repeats, (TG)n / poly-T microsatellites (huge 3-letter buckets)."""
    rng = np.random.default_rng(seed)
    out = []
    fam = random_seq(rng, 300, gc)
    for ci, L in enumerate(chr_lens):
        s = random_seq(rng, L, gc).copy()
        for _ in range(repeats):  # dispersed repeat family, ~8 % divergence
            ln = int(rng.integers(120, 300))
            pos = int(rng.integers(0, max(1, L - ln)))
            cp = fam[:ln].copy()
            mut = rng.random(ln) < 0.08
            cp[mut] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(mut.sum()))]
            if rng.random() < 0.5:
                cp = np.frombuffer(cp.tobytes().translate(COMP)[::-1], dtype=np.uint8)
            s[pos:pos + ln] = cp
        for k in range(microsats):
            ln = int(rng.integers(40, 400))
            pos = int(rng.integers(0, max(1, L - ln)))
            unit = [b"TG", b"T", b"CA", b"A", b"TTG"][k % 5]
            s[pos:pos + ln] = np.frombuffer((unit * (ln // len(unit) + 1))[:ln], dtype=np.uint8)
        for _ in range(cpg_sites):  # RRBS: sprinkle digestion sites
            pos = int(rng.integers(0, max(1, L - 8)))
            s[pos:pos + len(digest)] = np.frombuffer(digest.encode(), dtype=np.uint8)
        for _ in range(lower):
            ln = int(rng.integers(50, 2000))
            pos = int(rng.integers(0, max(1, L - ln)))
            s[pos:pos + ln] |= 0x20
        for _ in range(iupac):
            pos = int(rng.integers(0, L))
            s[pos] = b"RYMKSW"[int(rng.integers(0, 6))]
        for k in range(n_runs):
            ln = int(rng.integers(1, 3000)) if k % 2 else int(rng.integers(1, 6))
            pos = int(rng.integers(0, max(1, L - ln)))
            s[pos:pos + ln] = ord("N") if k % 3 else ord("n")
            if k % 2 == 0 and pos + ln + 20 + 4 < L:  # a short island between two N runs (< 30 nt -> not indexed)
                s[pos + ln + 20:pos + ln + 24] = ord("N")
        if ci == 0 and L > 200:
            s[:37] = ord("N")  # leading Ns
        out.append((f"chr{ci + 1}", s.tobytes().decode()))
    return out


def write_fasta(path, genome, width=70, extra_header=" some description"):
    with open(path, "w") as f:
        for name, s in genome:
            f.write(f">{name}{extra_header}\n")
            for i in range(0, len(s), width):
                f.write(s[i:i + width] + "\n")


def fasta_text(genome, width=70):
    parts = []
    for name, s in genome:
        parts.append(f">{name} desc\n")
        parts.extend(s[i:i + width] + "\n" for i in range(0, len(s), width))
    return "".join(parts)


def bs_convert(frag: str, rng, conv_nonCpG=0.995, conv_CpG=0.25):
    b = bytearray(frag.upper().encode())
    n = len(b)
    r = rng.random(n)
    for i in range(n):
        if b[i] == 67:  # C
            cpg = i + 1 < n and b[i + 1] == 71
            if r[i] < (conv_CpG if cpg else conv_nonCpG):
                b[i] = 84
    return b.decode()


def mutate(seq: str, rng, sub_rate=0.005, n_rate=0.0, max_subs=None):
    b = bytearray(seq.encode())
    n = len(b)
    r = rng.random(n)
    subs = np.nonzero(r < sub_rate)[0]
    if max_subs is not None:
        subs = subs[:max_subs]
    for i in subs:
        b[i] = b"ACGT"[(b"ACGT".find(bytes([b[i]])) + int(rng.integers(1, 4))) % 4] if bytes([b[i]]) in b"ACGT" else b[i]
    if n_rate > 0:
        for i in np.nonzero(rng.random(n) < n_rate)[0]:
            b[i] = ord("N")
    return b.decode()


def _sample_fragment(genome, rng, length):
    while True:
        ci = int(rng.integers(0, len(genome)))
        name, s = genome[ci]
        if len(s) < length + 2:
            continue
        pos = int(rng.integers(0, len(s) - length))
        frag = s[pos:pos + length]
        if frag.upper().count("N") > length // 2:
            continue
        return ci, pos, frag


def make_se_reads(genome, n, length, seed=1, sub_rate=0.005, n_rate=0.001, strands=("++", "-+"), var_len=False,
                  junk_frac=0.02, qual_tail=False, adapter=None):
    """returns list of dict(name, seq, qual, chr, pos, strand)"""
    rng = np.random.default_rng(seed)
    reads = []
    for i in range(n):
        L = int(rng.integers(max(20, length // 2), length + 1)) if var_len else length
        if rng.random() < junk_frac:
            seq = random_seq(rng, L).tobytes().decode()
            reads.append(dict(name=f"r{i}_junk", seq=seq, qual="I" * L, chr=-1, pos=-1, strand="*"))
            continue
        ci, pos, frag = _sample_fragment(genome, rng, L)
        strand = strands[int(rng.integers(0, len(strands)))]
        if strand[0] == "+":
            conv = bs_convert(frag, rng)
        else:
            conv = bs_convert(revcomp(frag.upper()), rng)
        seq = conv if strand[1] == "+" else revcomp(conv)
        seq = mutate(seq, rng, sub_rate, n_rate)
        qual = "I" * L
        if adapter is not None and rng.random() < 0.3:
            ins = int(rng.integers(min(30, L - 1), L))  # (the same draw for L > 30)
            seq = (seq[:ins] + adapter + random_seq(rng, L).tobytes().decode())[:L]
        if qual_tail:
            t = int(rng.integers(0, 60))
            if t:
                q = bytearray(qual.encode())
                q[L - t:] = bytes(int(x) for x in rng.integers(35, 49, t))
                qual = q.decode()
        reads.append(dict(name=f"r{i}_{genome[ci][0]}_{pos + 1}_{strand}", seq=seq, qual=qual, chr=ci, pos=pos, strand=strand))
    return reads


def make_pe_reads(genome, n, length, seed=1, ins_mean=300, ins_sd=50, ins_min=50, ins_max=480, sub_rate=0.005,
                  n_rate=0.001, junk_frac=0.02, qual_tail=False, adapter=None, var_len=False):
    """mate 1 from ++ / -+, mate 2 the reverse complement of the far end (+- / --)"""
    rng = np.random.default_rng(seed)
    pairs = []
    for i in range(n):
        ins = int(np.clip(rng.normal(ins_mean, ins_sd), ins_min, ins_max))
        ci, pos, frag = _sample_fragment(genome, rng, ins)
        watson = rng.random() < 0.5
        conv = bs_convert(frag if watson else revcomp(frag.upper()), rng)
        L1 = int(rng.integers(max(20, length // 2), length + 1)) if var_len else length
        L2 = int(rng.integers(max(20, length // 2), length + 1)) if var_len else length
        filler = random_seq(rng, 2 * length).tobytes().decode()
        ad = adapter if adapter is not None else ""
        m1 = (conv + ad + filler)[:L1]
        m2 = (revcomp(conv) + ad + filler)[:L2]
        if rng.random() < junk_frac:
            m2 = random_seq(rng, L2).tobytes().decode()
        m1 = mutate(m1, rng, sub_rate, n_rate)
        m2 = mutate(m2, rng, sub_rate, n_rate)
        q1, q2 = "I" * L1, "I" * L2
        if qual_tail:
            for which in (0, 1):
                L = (L1, L2)[which]
                t = int(rng.integers(0, 60))
                if t:
                    q = bytearray(b"I" * L)
                    q[L - t:] = bytes(int(x) for x in rng.integers(35, 49, t))
                    if which == 0:
                        q1 = q.decode()
                    else:
                        q2 = q.decode()
        pairs.append(dict(name=f"p{i}_{genome[ci][0]}_{pos + 1}_{'W' if watson else 'C'}_{ins}", seq1=m1, qual1=q1, seq2=m2,
                          qual2=q2, chr=ci, pos=pos, ins=ins, watson=watson))
    return pairs


def make_rrbs_reads(genome, n, length, seed=1, digest="CCGG", digest_pos=1, sub_rate=0.005, max_frag=220, min_frag=40):
    """reads starting at digestion sites (C-CGG): fragment = [site_i + pos, site_j + pos + ...)"""
    rng = np.random.default_rng(seed)
    reads = []
    sites = []
    for ci, (name, s) in enumerate(genome):
        u = s.upper()
        p = u.find(digest)
        lst = []
        while p >= 0:
            lst.append(p + digest_pos)
            p = u.find(digest, p + 1)
        sites.append(lst)
    tries = 0
    while len(reads) < n and tries < 50 * n:
        tries += 1
        ci = int(rng.integers(0, len(genome)))
        lst = sites[ci]
        if len(lst) < 2:
            continue
        k = int(rng.integers(0, len(lst) - 1))
        a, b = lst[k], lst[k + 1] + len(digest) - 2 * digest_pos
        if not (min_frag <= b - a <= max_frag):
            continue
        frag = genome[ci][1][a:b]
        watson = rng.random() < 0.5
        conv = bs_convert(frag if watson else revcomp(frag.upper()), rng)
        seq = mutate(conv[:length], rng, sub_rate, 0.0)
        if len(seq) < 20:
            continue
        reads.append(dict(name=f"rr{len(reads)}_{genome[ci][0]}_{a + 1}_{'W' if watson else 'C'}", seq=seq, qual="I" * len(seq),
                          chr=ci, pos=a, strand="++" if watson else "-+"))
    return reads


def write_fastq(path, reads, seq_key="seq", qual_key="qual"):
    with open(path, "w") as f:
        for r in reads:
            f.write(f"@{r['name']}\n{r[seq_key]}\n+\n{r[qual_key]}\n")


def c1_full_inputs(tmp):
    """BASELINE.json's configs[0] at its stated size: 1 Mb genome FASTA + 10 000 single-end 36 bp reads (200 of them of
    varying length, so that the planner-state leak of DESIGN.md §4 occurs); returns (fasta path, fastq path, sha256 of both
    texts).  tests/golden/c1_full.json.gz holds the real binary's SAM for exactly these files."""
    import hashlib
    import os
    g = make_genome(seed=1, chr_lens=(1_000_000,), gc=0.51)
    fa = os.path.join(tmp, "c1_genome.fa")
    write_fasta(fa, g)
    reads = make_se_reads(g, 9_800, 36, seed=101, sub_rate=0.01, strands=("++", "-+"))
    reads += make_se_reads(g, 200, 36, seed=102, sub_rate=0.04, strands=("++", "-+"), var_len=True)
    fq = os.path.join(tmp, "c1_reads.fq")
    write_fastq(fq, reads)
    h = hashlib.sha256(open(fa, "rb").read() + open(fq, "rb").read()).hexdigest()
    return fa, fq, h
