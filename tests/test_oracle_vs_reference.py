"""Live pinning of the oracle against the REAL reference objects (oracle/_ref/libbsmapref.so, built from
/root/reference by `make -C oracle ref`).  Skipped where the reference is not present (GPU box)."""
import os

import numpy as np
import pytest

import bsx_testdata as td
from oracle import ref_ffi as R

pytestmark = pytest.mark.skipif(not (R.available() or os.path.isdir(R.REFERENCE_DIR)), reason="reference not available")

ADAPTER = "AGATCGGAAGAGC"


@pytest.fixture(scope="module")
def genome(tmp_path_factory):
    if not R.available():
        R.build()
    d = tmp_path_factory.mktemp("ref")
    g = td.make_genome(seed=1, chr_lens=(200_000, 90_000, 30_017), gc=0.51)
    fa = str(d / "g.fa")
    td.write_fasta(fa, g)
    gr = td.make_genome(seed=4, chr_lens=(250_000, 80_000), gc=0.55, cpg_sites=1200)
    far = str(d / "gr.fa")
    td.write_fasta(far, gr)
    return g, fa, gr, far


def _state_eq(rs, os_, nclass, tag):
    for f in ("filtered", "len", "raw_len"):
        assert getattr(rs, f) == getattr(os_, f), (tag, f)
    if rs.filtered:
        return
    for f in ("read_max_snp_num", "seedseg_num", "flag_chain", "cflag_chain", "snp_thres"):
        assert getattr(rs, f) == getattr(os_, f), (tag, f)
    n = rs.seedseg_num
    if rs.flag_chain:
        assert list(rs.seed_start_array)[:n] == list(os_.seed_start_array)[:n], tag
        assert list(rs.seedindex)[:n] == list(os_.seedindex)[:n] and list(rs.seedcount)[:n] == list(os_.seedcount)[:n], tag
    if rs.cflag_chain:
        assert list(rs.cseed_start_array)[:n] == list(os_.cseed_start_array)[:n], tag
        assert list(rs.cseedindex)[:n] == list(os_.cseedindex)[:n], tag
    assert list(rs.n_hit)[:nclass] == list(os_.n_hit)[:nclass] and list(rs.n_chit)[:nclass] == list(os_.n_chit)[:nclass], tag


SE_CASES = [
    dict(kw=dict(s=12, v=2, I=4, S=1, r=1, out_sam=1), length=36, sub=0.02, strands=("++", "-+")),
    dict(kw=dict(s=16, v=4, I=4, S=1, r=1, n=1, out_sam=1), length=100, sub=0.01),
    dict(kw=dict(s=16, v=4, I=4, S=7, r=0, n=0, out_sam=1), length=100, sub=0.01),
    dict(kw=dict(s=16, v=6, I=4, S=3, r=1, n=1, out_sam=1, q=20, A=[ADAPTER]), length=144, sub=0.01, var=True, trim=True),
    dict(kw=dict(s=14, v=5, I=2, S=3, r=1, n=1, out_sam=1, w=5), length=144, sub=0.01, var=True),
    dict(kw=dict(s=10, v=3, I=1, S=3, r=0, n=1, out_sam=1, w=3), length=144, sub=0.01, var=True),
    dict(kw=dict(s=9, v=15, I=16, S=3, r=1, n=1, out_sam=0, w=1000, M="GA"), length=80, sub=0.02, n=300),
]


@pytest.mark.parametrize("case", SE_CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c["kw"].items() if k not in ("A", "out_sam")))
def test_se(case, genome, oracle):
    g, fa, _, _ = genome
    kw = case["kw"]
    ref = R.Reference(fa, **kw)
    p = oracle.make_params(**kw)
    o = oracle.OracleRef(p, fasta_path=fa)
    assert np.array_equal(ref.refcat()[400:-400], o.refcat()[400:-400]) and np.array_equal(ref.crefcat()[400:-400], o.crefcat()[400:-400])
    assert np.array_equal(ref.blocks(), o.blocks()) and np.array_equal(ref.anchor(), o.anchor())
    if kw["s"] <= 12:
        off, nf, ent = ref.csr()
        assert np.array_equal(off, o.bucket_off()) and np.array_equal(nf, o.bucket_nfwd()) and np.array_equal(ent, o.entries())
    reads = case.get("reads") or td.make_se_reads(g, case.get("n", 1500), case["length"], seed=2, sub_rate=case["sub"], var_len=case.get("var", False),
                                                  strands=case.get("strands", ("++", "-+", "+-", "--")), qual_tail=case.get("trim", False),
                                                  adapter=ADAPTER if case.get("trim") else None)
    al = oracle.OracleAligner(o, leak_mode=1)
    nclass = kw["v"] + 1
    names = o.names()
    for i, r in enumerate(reads):
        rs, line = ref.se(i, r["name"], r["seq"], r["qual"])
        os_ = al.se(i, r["seq"], r["qual"])
        _state_eq(rs, os_, nclass, (i, r["name"]))
        if rs.filtered:
            continue
        for w in range(nclass):
            for orient in (0, 1):
                n = (rs.n_chit if orient else rs.n_hit)[w]
                assert ref.se_hits(orient, w, n) == al.se_hits(orient, w, n), (i, w, orient)
        f = line.split("\t")
        if kw.get("out_sam") and line and f[2] != "*":
            assert f[2] == names[os_.chr >> 1] and int(f[3]) == os_.loc + 1 and f[11] == "NM:i:%d" % os_.best_class
            assert "ZS:Z:" + "+-"[os_.chr & 1] + "+-"[os_.chain] in line
    al.free()
    o.free()


PE_CASES = [
    dict(kw=dict(s=16, v=6, I=4, S=1, r=1, m=28, x=500, out_sam=1), length=150),
    dict(kw=dict(s=16, v=6, I=4, S=1, r=0, m=28, x=500, out_sam=1), length=150),
    dict(kw=dict(s=16, v=6, I=4, S=1, r=1, m=28, x=500, out_sam=1, q=20, A=[ADAPTER]), length=150, trim=True),
    dict(kw=dict(s=12, v=3, I=2, S=2, r=1, n=1, m=0, x=300, out_sam=1, w=4), length=150, trim=True),
]


@pytest.mark.parametrize("case", PE_CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c["kw"].items() if k not in ("A", "out_sam")))
def test_pe(case, genome, oracle):
    g, fa, _, _ = genome
    kw = dict(case["kw"], pairend=1)
    ref = R.Reference(fa, **kw)
    p = oracle.make_params(**kw)
    o = oracle.OracleRef(p, fasta_path=fa)
    trim = case.get("trim", False)
    pairs = td.make_pe_reads(g, 1000, case["length"], seed=6, sub_rate=0.015, qual_tail=trim, adapter=ADAPTER if trim else None,
                             var_len=trim, ins_min=20 if trim else 50, ins_mean=200 if trim else 300, ins_sd=100 if trim else 50)
    al = oracle.OracleAligner(o, leak_mode=1)
    nclass = kw["v"] + 1
    for i, r in enumerate(pairs):
        rs, l1, l2 = ref.pe(i, r["name"] + "/1", r["seq1"], r["qual1"], r["name"] + "/2", r["seq2"], r["qual2"])
        os_ = al.pe(i, r["seq1"], r["seq2"], r["qual1"], r["qual2"])
        assert (rs.paired, rs.tmp, list(rs.n_pairs)) == (os_.paired, os_.tmp, list(os_.n_pairs)), i
        _state_eq(rs.a, os_.a, nclass, (i, "a"))
        _state_eq(rs.b, os_.b, nclass, (i, "b"))
        for mate, x in enumerate((rs.a, rs.b)):
            if x.filtered:
                continue
            for w in range(nclass):
                for orient in (0, 1):
                    n = (x.n_chit if orient else x.n_hit)[w]
                    assert ref.pe_hits(mate, orient, w, n) == al.pe_hits(mate, orient, w, n), (i, mate, w, orient)
        for w in range(2 * nclass - 1):
            assert ref.pe_pairs(w, rs.n_pairs[w]) == al.pe_pairs(w, rs.n_pairs[w]), (i, w)
    al.free()
    o.free()


@pytest.mark.parametrize("kw", [dict(D="C-CGG", v=4, S=1, r=1, out_sam=1), dict(D="C-CGG", v=2, S=1, r=0, n=1, out_sam=1, A=[ADAPTER])],
                         ids=["rrbs-v4", "rrbs-n1-r0"])
def test_rrbs(kw, genome, oracle):
    _, _, g, fa = genome
    ref = R.Reference(fa, **kw)
    p = oracle.make_params(**kw)
    o = oracle.OracleRef(p, fasta_path=fa)
    off, ent = ref.rrbs_csr()
    assert np.array_equal(off, o.bucket_off()) and np.array_equal(ent, o.rrbs_entries())
    for c in range(2):
        assert np.array_equal(ref.sites(c), o.sites(c))
    al = oracle.OracleAligner(o, leak_mode=1)
    nclass = kw["v"] + 1
    reads = td.make_rrbs_reads(g, 1200, 75, seed=3) + td.make_se_reads(g, 200, 75, seed=9, var_len=True)
    for i, r in enumerate(reads):
        rs, line = ref.se(i, r["name"], r["seq"], r["qual"])
        os_ = al.se(i, r["seq"], r["qual"])
        for f in ("filtered", "len"):
            assert getattr(rs, f) == getattr(os_, f)
        if rs.filtered:
            continue
        assert list(rs.n_hit)[:nclass] == list(os_.n_hit)[:nclass] and list(rs.n_chit)[:nclass] == list(os_.n_chit)[:nclass], i
        for w in range(nclass):
            for orient in (0, 1):
                n = (rs.n_chit if orient else rs.n_hit)[w]
                assert ref.se_hits(orient, w, n) == al.se_hits(orient, w, n)
    al.free()
    o.free()


@pytest.mark.parametrize("q", [5, 20])
def test_quality_string_longer_or_shorter_than_the_read(q, genome, oracle):
    """a malformed FASTQ record may carry more or fewer quality characters than bases: the reference keeps both lengths, and its
    TrimLowQual scans the whole quality string (align.cpp:69-78) — a good character beyond the last base keeps the read untrimmed, a
    short string cuts the read to its length.  (At the C ABI a batch has one offset array for bases and qualities, so the device path
    never sees such a record; the command line cuts or pads the quality string to the read)"""
    g, fa, _, _ = genome
    kw = dict(s=14, v=4, I=1, S=54, r=0, n=1, w=50, out_sam=1, q=q)
    rng = np.random.default_rng(5)
    reads = td.make_se_reads(g, 120, 60, seed=9, sub_rate=0.004, strands=("++", "-+"))
    for i, r in enumerate(reads):
        L = len(r["seq"])
        tail = "".join(chr(int(x)) for x in rng.integers(35, 49, 30))
        if i % 3 == 0:
            r["qual"] = "I" * (L - 25) + tail[:25] + tail            # longer than the read, low tail with a few good characters beyond it
        elif i % 3 == 1:
            r["qual"] = "I" * (L - 20)                               # shorter than the read
        else:
            r["qual"] = "I" * (L - 30) + tail                        # same length, low tail
    test_se(dict(kw=kw, reads=reads), genome, oracle)


def _random_case(seed):
    """a seeded draw from the option space of the command line (main.cpp:234-289) — the same space the GPU path is
    fuzzed over in tests/test_gpu_parity.py"""
    import random
    rng = random.Random(seed)
    kw = dict(s=rng.randint(9, 16), v=rng.choice([0, 1, 2, 3, 4, 5, 6, 8, 11]), I=rng.choice([1, 2, 3, 4, 4, 5, 8]), S=rng.randint(1, 99),
              r=rng.choice([0, 1, 1]), n=rng.choice([0, 1]), w=rng.choice([1, 2, 7, 50, 1000]), f=rng.choice([0, 2, 5]),
              L=rng.choice([144, 144, 100, 61]), out_sam=1)
    if rng.random() < 0.3:
        kw["M"] = rng.choice(["GA", "CT", "AG", "TG"])
    trim = rng.random() < 0.4
    if trim:
        kw.update(q=rng.choice([5, 20, 30]), A=[ADAPTER])
    return kw, trim, rng


@pytest.mark.parametrize("seed", list(range(301, 333)))
def test_se_random_options(seed, genome, oracle):
    kw, trim, rng = _random_case(seed)
    test_se(dict(kw=kw, length=rng.choice([150, 120, 80, 40]), sub=0.006, var=rng.random() < 0.5, trim=trim, n=500), genome, oracle)


@pytest.mark.parametrize("seed", list(range(401, 421)))
def test_pe_random_options(seed, genome, oracle):
    kw, trim, rng = _random_case(seed)
    kw.pop("n")
    kw.update(m=rng.choice([0, 28, 120]), x=rng.choice([250, 500, 900]))
    test_pe(dict(kw=kw, length=rng.choice([150, 120, 80]), trim=trim), genome, oracle)


@pytest.mark.parametrize("seed", list(range(501, 515)))
def test_rrbs_random_options(seed, genome, oracle):
    import random
    rng = random.Random(seed)
    kw = dict(D=rng.choice(["C-CGG", "C-CGG", "CCG-G", "-CCGG"]), v=rng.choice([0, 1, 2, 3, 5]), S=rng.randint(1, 90), r=rng.choice([0, 1, 1]),
              w=rng.choice([1, 3, 100, 1000]), m=rng.choice([20, 40, 80]), x=rng.choice([150, 220, 400]), L=rng.choice([144, 75, 50]),
              n=rng.choice([0, 1]), out_sam=1)
    test_rrbs(kw, genome, oracle)
