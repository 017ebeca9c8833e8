"""Parity of the HIP path (through the C ABI of libbsx.so) with the oracle and the golden vectors.  Needs an MI355X.

Bar: bit-exact — packed reference words, block list, every index bucket, the planner state, every hit list, every
pair list, the chosen hit / pair, and the work counters that feed the roofline numerator.
"""
import os

import numpy as np
import pytest

import bsmap_amd as B
import bsx_testdata as td
import golden_util as G

EXTRA_FUZZ = int(os.environ.get("BSX_EXTRA_FUZZ", "0"))   # N more seeded option draws in every fuzz test below, for a one-off soak outside the suite

pytestmark = pytest.mark.gpu

ADAPTER = "AGATCGGAAGAGC"


def _counters_off():
    """BSX_WORK_COUNTERS=0: batches are created with the work counters off (include/bsx.h) — records are compared, the counters are not"""
    return os.environ.get("BSX_WORK_COUNTERS") == "0"


def _leaky(length, kw):
    """reads whose reference result depends on the previous read (SURVEY §7 hard part 2): (len-I+1) % S == 0"""
    if "D" in kw:
        return False
    return (length - kw.get("I", 4) + 1) % kw.get("s", 16) == 0


def _cmp_read(o, h, cc, kw, tag):
    """o: oracle ReadResult, h: bsx_hit record, cc: class counts"""
    nclass = kw["v"] + 1
    assert bool(o.filtered) == bool(h["flags"] & B.F_FILTERED), tag
    assert o.len == h["len"] and o.raw_len == h["raw_len"], tag
    if o.filtered:
        return
    assert o.read_max_snp_num == h["max_snp"] and o.seedseg_num == h["seedseg"], tag
    assert list(o.n_hit)[:nclass] == list(cc["n_hit"][:nclass]), (tag, list(o.n_hit)[:nclass], list(cc["n_hit"][:nclass]))
    assert list(o.n_chit)[:nclass] == list(cc["n_chit"][:nclass]), (tag, list(o.n_chit)[:nclass], list(cc["n_chit"][:nclass]))


def _cmp_pick(o, h, tag):
    assert max(o.n_best, 0) == h["n_best"], tag
    if o.n_best > 0:
        assert (o.best_class, o.chr, o.loc, o.chain) == (h["best_class"], h["chr"], h["loc"], (h["flags"] >> 1) & 1), tag
    else:
        assert h["best_class"] == -1, tag


@pytest.fixture(scope="module", params=G.CONFIGS)
def case(request, oracle):
    meta, arr, fasta = G.load(request.param)
    kw = meta["kw"]
    op = oracle.make_params(**kw)
    oref = oracle.OracleRef(op, fasta_path=fasta)
    gp = B.make_params(**kw)
    gref = B.RefSeq(gp).Run_ConvertBinseq(fasta_path=fasta).CreateIndex()
    yield meta, arr, oref, gref, oracle
    gref.close()
    oref.free()


def test_reference_and_index(case):
    meta, arr, oref, gref, O = case
    a, s, r = gref.info()
    assert np.array_equal(a, arr["anchor"]) and np.array_equal(s, arr["chr_size"]) and np.array_equal(r, arr["rc_offset"])
    assert np.array_equal(gref.blocks(), arr["blocks"])
    f, c = gref.words()
    assert np.array_equal(f[400:-400], arr["refcat"]) and np.array_equal(c[400:-400], arr["crefcat"])
    assert gref.names() == oref.names()
    off, nf, ent = gref.index()
    if "D" in meta["kw"]:
        keys, n, _ = G.sparse_index(off)
        assert np.array_equal(keys, arr["idx_keys"]) and np.array_equal(n, arr["idx_n"]) and np.array_equal(ent, arr["idx_entries"])
        assert np.array_equal(gref.sites(0), arr["sites0"]) and np.array_equal(gref.sites(1), arr["sites1"])
    else:
        keys, n, nfk = G.sparse_index(off, nf)
        assert np.array_equal(keys, arr["idx_keys"]) and np.array_equal(n, arr["idx_n"]) and np.array_equal(nfk, arr["idx_nfwd"])
        assert np.array_equal(ent, arr["idx_entries"])
        assert np.array_equal(off, oref.bucket_off()) and np.array_equal(nf, oref.bucket_nfwd())


def test_alignment_vs_oracle_and_golden(case):
    meta, arr, oref, gref, O = case
    kw = meta["kw"]
    nclass = kw["v"] + 1
    reads = meta["reads"]
    al = O.OracleAligner(oref, leak_mode=0)
    if meta["kind"] == "se":
        sa = B.SingleAlign(gref, len(reads), debug=True)
        sa.ImportBatchReads([r["seq"] for r in reads], [r["qual"] for r in reads]).Do_Batch()
        hits, cc = sa.results()
        for i, r in enumerate(reads):
            o = al.se(i, r["seq"], r["qual"])
            _cmp_read(o, hits[i], cc[i], kw, (meta["config"], i))
            if o.filtered:
                continue
            _cmp_pick(o, hits[i], i)
            st, od = sa.debug_plan(i)
            n = o.seedseg_num
            if o.flag_chain:
                assert list(st[0][:n]) == list(o.seed_start_array)[:n] and list(od[0][:n]) == list(o.seedindex)[:n], i
            if o.cflag_chain:
                assert list(st[1][:n]) == list(o.cseed_start_array)[:n] and list(od[1][:n]) == list(o.cseedindex)[:n], i
            for w in range(nclass):
                for orient in (0, 1):
                    nn = (o.n_chit if orient else o.n_hit)[w]
                    assert sa.debug_hits(i, 0, orient, w) == al.se_hits(orient, w, nn), (i, w, orient)
            # golden record from the real reference (skip the reads whose reference result is call-order dependent)
            e = meta["expected"][i]
            if not e["filtered"] and not _leaky(e["len"], kw):
                assert e["n_hit"][:nclass] == list(cc[i]["n_hit"][:nclass]) and e["n_chit"][:nclass] == list(cc[i]["n_chit"][:nclass]), i
                for w in range(nclass):
                    for orient in (0, 1):
                        assert [tuple(x) for x in e["hits"][w][orient]] == sa.debug_hits(i, 0, orient, w), (i, w, orient)
        c = sa.counters()
        assert _counters_off() or [int(x) for x in c[:4]] == al.counters(), (c, al.counters())
        assert int(c[4]) == len(reads)
        sa.close()
    else:
        pa = B.PairAlign(gref, len(reads), debug=True)
        pa.ImportBatchReads([r["seq1"] for r in reads], [r["seq2"] for r in reads], [r["qual1"] for r in reads], [r["qual2"] for r in reads]).Do_Batch()
        out, ca, cb, npairs = pa.results()
        for i, r in enumerate(reads):
            o = al.pe(i, r["seq1"], r["seq2"], r["qual1"], r["qual2"])
            g = out[i]
            _cmp_read(o.a, g["a"], ca[i], kw, (i, "a"))
            _cmp_read(o.b, g["b"], cb[i], kw, (i, "b"))
            assert o.paired == g["paired"], (i, o.paired, g["paired"])
            assert list(o.n_pairs)[:2 * nclass - 1] == list(npairs[i][:2 * nclass - 1]), i
            for mate, om in enumerate((o.a, o.b)):
                if om.filtered:
                    continue
                for w in range(nclass):
                    for orient in (0, 1):
                        nn = (om.n_chit if orient else om.n_hit)[w]
                        assert pa.debug_hits(i, mate, orient, w) == al.pe_hits(mate, orient, w, nn), (i, mate, w, orient)
            for w in range(2 * nclass - 1):
                assert pa.debug_pairs(i, w) == al.pe_pairs(w, o.n_pairs[w]), (i, w)
            assert bool(g["unpaired_out"]) == bool(o.tmp == 1 or o.paired == 0), i
            if not g["unpaired_out"]:
                pk = o.pick
                assert (pk.chain, pk.na, pk.nb, pk.insert, pk.a.chr, pk.a.loc, pk.b.chr, pk.b.loc) == \
                       (g["chain"], g["na"], g["nb"], g["insert"], g["a_chr"], g["a_loc"], g["b_chr"], g["b_loc"]), i
                assert (o.pair_class, o.pair_n) == (g["pair_class"], g["n_pairs"]), i
            else:
                _cmp_pick(o.a, g["a"], (i, "a")) if not o.a.filtered else None
                _cmp_pick(o.b, g["b"], (i, "b")) if not o.b.filtered else None
            e = meta["expected"][i]
            if not e["a"]["filtered"] and not e["b"]["filtered"] and not _leaky(e["a"]["len"], kw) and not _leaky(e["b"]["len"], kw):
                assert e["paired"] == g["paired"] and e["n_pairs"][:2 * nclass - 1] == list(npairs[i][:2 * nclass - 1]), i
                for w, pl in enumerate(e["pairs"]):
                    assert [tuple(x) for x in pl] == pa.debug_pairs(i, w), (i, w)
        c = pa.counters()
        assert _counters_off() or [int(x) for x in c[:4]] == al.counters(), (c, al.counters())
        pa.close()
    al.free()


# ---- seeded larger cases: edge conditions the golden sets are too small for --------------------------------
EDGE = [
    ("se_n1_var", dict(s=16, v=6, I=4, S=3, r=1, n=1, q=20, A=[ADAPTER]), dict(kind="se", n=4000, length=144, var=True, trim=True)),
    ("se_w5", dict(s=14, v=5, I=2, S=3, r=1, n=1, w=5), dict(kind="se", n=3000, length=144, var=True)),
    ("se_r0_w3", dict(s=10, v=3, I=1, S=3, r=0, n=1, w=3), dict(kind="se", n=3000, length=120, var=True)),
    ("se_v15_I16_GA", dict(s=9, v=15, I=16, S=3, r=1, n=1, M="GA"), dict(kind="se", n=600, length=80)),
    ("se_s12_c1", dict(s=12, v=2, I=4, S=1, r=1), dict(kind="se", n=10000, length=36, sub=0.02)),
    ("pe_c3", dict(s=16, v=6, I=4, S=1, r=1, m=28, x=500), dict(kind="pe", n=3000, length=150)),
    ("pe_trim_w4", dict(s=12, v=3, I=2, S=2, r=1, n=1, m=0, x=300, w=4), dict(kind="pe", n=2000, length=150, trim=True)),
    ("pe_r0_q", dict(s=16, v=6, I=4, S=1, r=0, m=28, x=500, q=20, A=[ADAPTER]), dict(kind="pe", n=2000, length=150, trim=True)),
]


@pytest.fixture(scope="module")
def edge_genome(tmp_path_factory):
    g = td.make_genome(seed=1, chr_lens=(300_000, 150_000, 50_017), gc=0.51, microsats=25, repeats=60)
    fa = str(tmp_path_factory.mktemp("edge") / "g.fa")
    td.write_fasta(fa, g)
    return g, fa


@pytest.mark.parametrize("name,kw,spec", EDGE, ids=[e[0] for e in EDGE])
def test_edge_cases_vs_oracle(name, kw, spec, edge_genome, oracle):
    g, fa = edge_genome
    if spec["kind"] == "pe":
        kw = dict(kw, pairend=1)
    op = oracle.make_params(**kw)
    oref = oracle.OracleRef(op, fasta_path=fa)
    gref = B.RefSeq(B.make_params(**kw)).Run_ConvertBinseq(fasta_path=fa).CreateIndex()
    off, nf, ent = gref.index()
    assert np.array_equal(off, oref.bucket_off()) and np.array_equal(nf, oref.bucket_nfwd()) and np.array_equal(ent, oref.entries())
    trim = spec.get("trim", False)
    if spec["kind"] == "se":
        reads = td.make_se_reads(g, spec["n"], spec["length"], seed=7, sub_rate=spec.get("sub", 0.01), var_len=spec.get("var", False),
                                 strands=("++", "-+", "+-", "--"), qual_tail=trim, adapter=ADAPTER if trim else None)
        seqs, quals = [r["seq"] for r in reads], [r["qual"] for r in reads]
        sbuf, soff = oracle.pack_reads(seqs)
        qbuf, _ = oracle.pack_reads(quals)
        ores, ocnt = oracle.se_batch(oref, sbuf, soff, qbuf, threads=4)
        sa = B.SingleAlign(gref, len(reads))
        sa.ImportBatchReads((sbuf, soff), qbuf).Do_Batch()
        hits, cc = sa.results()
        nclass = kw["v"] + 1
        assert np.array_equal(ores["filtered"] != 0, (hits["flags"] & 1) != 0)
        ok = ores["filtered"] == 0
        assert np.array_equal(ores["len"], hits["len"]) and np.array_equal(ores["n_hit"][ok][:, :nclass], cc["n_hit"][ok][:, :nclass])
        assert np.array_equal(ores["n_chit"][ok][:, :nclass], cc["n_chit"][ok][:, :nclass])
        assert np.array_equal(np.maximum(ores["n_best"][ok], 0), hits["n_best"][ok])
        has = ok & (ores["n_best"] > 0)
        for f in ("chr", "loc", "best_class"):
            assert np.array_equal(ores[f][has], hits[f][has]), f
        assert np.array_equal(ores["chain"][has], (hits["flags"][has] >> 1) & 1)
        assert _counters_off() or [int(x) for x in sa.counters()[:4]] == ocnt
        assert has.sum() > spec.get("min_frac", 0.2) * len(reads)
        sa.close()
    else:
        pairs = td.make_pe_reads(g, spec["n"], spec["length"], seed=8, sub_rate=0.015, qual_tail=trim, adapter=ADAPTER if trim else None,
                                 var_len=trim, ins_min=20 if trim else 50, ins_mean=200 if trim else 300, ins_sd=100 if trim else 50)
        s1, o1 = oracle.pack_reads([p["seq1"] for p in pairs])
        s2, o2 = oracle.pack_reads([p["seq2"] for p in pairs])
        q1, _ = oracle.pack_reads([p["qual1"] for p in pairs])
        q2, _ = oracle.pack_reads([p["qual2"] for p in pairs])
        ores, ocnt = oracle.pe_batch(oref, s1, o1, s2, o2, q1, q2, threads=4)
        pa = B.PairAlign(gref, len(pairs))
        pa.ImportBatchReads((s1, o1), (s2, o2), q1, q2).Do_Batch()
        out, ca, cb, npairs = pa.results()
        nclass = kw["v"] + 1
        assert np.array_equal(ores["paired"], out["paired"])
        assert np.array_equal(ores["n_pairs"][:, :2 * nclass - 1], npairs[:, :2 * nclass - 1])
        up = (ores["tmp"] == 1) | (ores["paired"] == 0)
        assert np.array_equal(up, out["unpaired_out"] != 0)
        pr = ~up
        for f, gname in (("chain", "chain"), ("na", "na"), ("nb", "nb"), ("insert", "insert"), ("a_chr", "a_chr"), ("a_loc", "a_loc"),
                         ("b_chr", "b_chr"), ("b_loc", "b_loc")):
            assert np.array_equal(ores["pick"][f][pr], out[gname][pr]), f
        for m, cnts in (("a", ca), ("b", cb)):
            ok = ores[m]["filtered"] == 0
            assert np.array_equal(ores[m]["n_hit"][ok][:, :nclass], cnts["n_hit"][ok][:, :nclass]), m
            assert np.array_equal(ores[m]["n_chit"][ok][:, :nclass], cnts["n_chit"][ok][:, :nclass]), m
            sel = up & ok & (ores[m]["n_best"] > 0)
            for f in ("chr", "loc", "best_class"):
                assert np.array_equal(ores[m][f][sel], out[m][f][sel]), (m, f)
        assert _counters_off() or [int(x) for x in pa.counters()[:4]] == ocnt
        assert pr.sum() > spec.get("min_frac", 0.2) * len(pairs)
        pa.close()
    gref.close()
    oref.free()


def _random_config(seed):
    """a seeded draw from the option space of the command line (main.cpp:234-289)"""
    import random
    rng = random.Random(seed)
    pe = rng.random() < 0.5
    kw = dict(s=rng.randint(9, 16), v=rng.choice([0, 1, 2, 3, 4, 5, 6, 8, 11]), I=rng.choice([1, 2, 3, 4, 4, 5, 8]), S=rng.randint(1, 99),
              r=rng.choice([0, 1, 1]), n=rng.choice([0, 1]), w=rng.choice([1, 2, 7, 50, 1000]), f=rng.choice([0, 2, 5]),
              L=rng.choice([144, 144, 100, 61]))
    if rng.random() < 0.3:
        kw["M"] = rng.choice(["GA", "CT", "AG", "TG"])
    trim = rng.random() < 0.4
    if trim:
        kw.update(q=rng.choice([5, 20, 30]), A=[ADAPTER])
    if pe:
        kw.update(m=rng.choice([0, 28, 120]), x=rng.choice([250, 500, 900]))
        kw.pop("n")  # (-n only changes single-end strand coverage in the paired driver: both mates are always placed)
    spec = dict(kind="pe" if pe else "se", n=1500 if pe else 2500, length=rng.choice([150, 120, 80]), var=rng.random() < 0.5, trim=trim,
                sub=0.004, min_frac=-1 if ("M" in kw or kw["v"] == 0) else 0.02)  # (the reads are C->T converted and carry errors: with another -M pair or -v 0 few align)
    return kw, spec


@pytest.mark.parametrize("seed", list(range(101, 221 + EXTRA_FUZZ)))
def test_random_option_combinations_vs_oracle(seed, edge_genome, oracle):
    """120 seeded draws from the option space (seed size, mismatches, interval, -w, -r, -n, -M, -f, -L, trimming, insert
    range, single / paired) through the same comparison as the edge cases: index, every count, every pick, work counters"""
    kw, spec = _random_config(seed)
    test_edge_cases_vs_oracle(f"fuzz{seed}", kw, spec, edge_genome, oracle)


@pytest.mark.parametrize("seed", list(range(1, 25 + EXTRA_FUZZ)))
def test_rrbs_random_options_vs_oracle(seed, oracle, tmp_path_factory):
    """RRBS mode (-D, site-anchored index, segment tags, fragment-size filter, short-fragment fix) under seeded option draws,
    single reads (with and without -n 1) and pairs; reads start at digestion sites of a CpG-rich genome"""
    import random
    rng = random.Random(500 + seed)
    if not hasattr(test_rrbs_random_options_vs_oracle, "_g"):
        g = td.make_genome(seed=11, chr_lens=(400_000, 120_000), gc=0.55, cpg_sites=900, repeats=30, microsats=10)
        fa = str(tmp_path_factory.mktemp("rrbs") / "g.fa")
        td.write_fasta(fa, g)
        test_rrbs_random_options_vs_oracle._g = (g, fa)
    g, fa = test_rrbs_random_options_vs_oracle._g
    pe = seed % 3 == 0
    kw = dict(D=rng.choice(["C-CGG", "C-CGG", "CCG-G", "-CCGG"]), v=rng.choice([0, 1, 2, 3, 5]), S=rng.randint(1, 90), r=rng.choice([0, 1, 1]), w=rng.choice([1, 3, 100, 1000]),
              m=rng.choice([20, 40, 80]), x=rng.choice([150, 220, 400]), L=rng.choice([144, 75, 50]))
    if not pe:
        kw["n"] = rng.choice([0, 1])
    else:
        kw["pairend"] = 1
    oref = oracle.OracleRef(oracle.make_params(**kw), fasta_path=fa)
    gref = B.RefSeq(B.make_params(**kw)).Run_ConvertBinseq(fasta_path=fa).CreateIndex()
    off, nf, ent = gref.index()
    assert np.array_equal(off, oref.bucket_off()) and np.array_equal(np.asarray(ent).reshape(-1, 2), oref.rrbs_entries())
    reads = td.make_rrbs_reads(g, 1500, rng.choice([75, 100, 36]), seed=seed, digest="CCGG", digest_pos=kw["D"].index("-"), sub_rate=0.006)
    nclass = kw["v"] + 1
    if not pe:
        sb, so = oracle.pack_reads([r["seq"] for r in reads])
        ores, ocnt = oracle.se_batch(oref, sb, so, threads=4)
        sa = B.SingleAlign(gref, len(reads))
        sa.ImportBatchReads((sb, so)).Do_Batch()
        hits, cc = sa.results()
        ok = ores["filtered"] == 0
        assert np.array_equal(ores["filtered"] != 0, (hits["flags"] & 1) != 0) and np.array_equal(ores["len"], hits["len"])
        assert np.array_equal(ores["n_hit"][ok][:, :nclass], cc["n_hit"][ok][:, :nclass]) and np.array_equal(ores["n_chit"][ok][:, :nclass], cc["n_chit"][ok][:, :nclass])
        has = ok & (ores["n_best"] > 0)
        for f in ("chr", "loc", "best_class"):
            assert np.array_equal(ores[f][has], hits[f][has]), f
        assert _counters_off() or [int(x) for x in sa.counters()[:4]] == ocnt
        if kw["D"] == "C-CGG" and kw["v"] >= 2 and kw["L"] >= 75:
            assert has.sum() > 300, has.sum()  # the reads really are site-anchored fragments
        test_rrbs_random_options_vs_oracle.last_heavy = sa.heavy_units()
        sa.close()
    else:
        # mates: the read and the reverse complement of the fragment's other end are not generated here; pair each read with
        # the next one — pairing mostly fails, which exercises the unpaired / short-fragment branches
        s1, o1 = oracle.pack_reads([r["seq"] for r in reads])
        s2, o2 = oracle.pack_reads([td.revcomp(r["seq"]) for r in reads[1:] + reads[:1]])
        ores, ocnt = oracle.pe_batch(oref, s1, o1, s2, o2, threads=4)
        pa = B.PairAlign(gref, len(reads))
        pa.ImportBatchReads((s1, o1), (s2, o2)).Do_Batch()
        out, ca, cb, npairs = pa.results()
        assert np.array_equal(ores["paired"], out["paired"]) and np.array_equal(ores["n_pairs"][:, :2 * nclass - 1], npairs[:, :2 * nclass - 1])
        up = (ores["tmp"] == 1) | (ores["paired"] == 0)
        assert np.array_equal(up, out["unpaired_out"] != 0)
        for f in ("chain", "na", "nb", "insert", "a_chr", "a_loc", "b_chr", "b_loc"):
            assert np.array_equal(ores["pick"][f][~up], out[f][~up]), f
        for m_, cnts in (("a", ca), ("b", cb)):
            ok = ores[m_]["filtered"] == 0
            assert np.array_equal(ores[m_]["n_hit"][ok][:, :nclass], cnts["n_hit"][ok][:, :nclass]) and np.array_equal(ores[m_]["n_chit"][ok][:, :nclass], cnts["n_chit"][ok][:, :nclass]), m_
            sel = up & ok & (ores[m_]["n_best"] > 0)
            for f in ("chr", "loc", "best_class"):
                assert np.array_equal(ores[m_][f][sel], out[m_][f][sel]), (m_, f)
        assert _counters_off() or [int(x) for x in pa.counters()[:4]] == ocnt
        test_rrbs_random_options_vs_oracle.last_heavy = pa.heavy_units()
        pa.close()
    gref.close()
    oref.free()


@pytest.mark.parametrize("seed", [1, 2, 3, 5, 6, 9, 12, 14, 17, 21])
def test_rrbs_through_the_heavy_pipeline(seed, oracle, tmp_path_factory):
    """the same RRBS draws with the heavy-unit threshold forced down to 2 candidates: buckets of {tag, loc} pairs go through the
    scan kernels (k_hscan_shared for runs of tasks over one window, segment / direction filter on the device, chromosome-local
    positions, per-entry strand), survivors through the 64-at-a-time replay with the per-lane fragment-size filter, every round
    runs (RRBS has no early stop) — hits, picks, pairs and the work counters must not change"""
    B.lib().bsx_set_heavy_threshold(2)
    try:
        test_rrbs_random_options_vs_oracle(seed, oracle, tmp_path_factory)
        assert test_rrbs_random_options_vs_oracle.last_heavy > 5
    finally:
        B.lib().bsx_set_heavy_threshold(0)


def test_empty_and_degenerate_inputs(edge_genome, oracle):
    """reads shorter than the seed, all-N reads, a read spanning a chromosome end, exact duplicates"""
    g, fa = edge_genome
    kw = dict(s=16, v=4, I=4, S=1, r=1, n=1)
    oref = oracle.OracleRef(oracle.make_params(**kw), fasta_path=fa)
    gref = B.RefSeq(B.make_params(**kw)).Run_ConvertBinseq(fasta_path=fa).CreateIndex()
    c0 = g[0][1].upper()
    tail = c0[-60:].replace("N", "A")
    seqs = ["ACGT", "A" * 15, "N" * 100, "ACGTN" * 20, tail + "ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT", c0[5000:5100], c0[5000:5100],
            td.revcomp(c0[7000:7100]), "T" * 144, "TG" * 72, c0[20000:20016], c0[30000:30200]]
    al = oracle.OracleAligner(oref, leak_mode=0)
    sa = B.SingleAlign(gref, len(seqs), debug=True)
    sa.ImportBatchReads(seqs).Do_Batch()
    hits, cc = sa.results()
    for i, s in enumerate(seqs):
        o = al.se(i, s)
        _cmp_read(o, hits[i], cc[i], kw, i)
        if not o.filtered:
            _cmp_pick(o, hits[i], i)
    assert hits[5]["n_best"] >= 1 and hits[5]["chr"] == 0 and hits[5]["loc"] == 5000
    sa.close()
    al.free()
    gref.close()
    oref.free()


@pytest.mark.parametrize("name,kw,spec", [e for e in EDGE if e[0] in ("se_n1_var", "se_w5", "se_r0_w3", "pe_c3", "pe_trim_w4")],
                         ids=lambda x: x if isinstance(x, str) else "")
def test_heavy_pipeline_forced_on_edge_cases(name, kw, spec, edge_genome, oracle):
    """the same cases with the heavy-unit threshold forced down to 48 candidates so that most units go through the
    heavy pipeline (windowed scans, event restarts, exact work accounting)"""
    B.lib().bsx_set_heavy_threshold(48)
    try:
        test_edge_cases_vs_oracle(name, kw, spec, edge_genome, oracle)
    finally:
        B.lib().bsx_set_heavy_threshold(0)


def test_heavy_pipeline_is_used(edge_genome, oracle):
    g, fa = edge_genome
    kw = dict(s=16, v=4, I=4, S=1, r=1, n=1)
    gref = B.RefSeq(B.make_params(**kw)).Run_ConvertBinseq(fasta_path=fa).CreateIndex()
    reads = td.make_se_reads(g, 2000, 100, seed=7, strands=("++", "-+", "+-", "--"))
    sa = B.SingleAlign(gref, len(reads))
    sa.ImportBatchReads([r["seq"] for r in reads])
    B.lib().bsx_set_heavy_threshold(48)
    try:
        sa.Do_Batch()
        h1, c1 = sa.results()
        n_heavy = sa.heavy_units()
    finally:
        B.lib().bsx_set_heavy_threshold(0)
    sa.Do_Batch()
    h2, c2 = sa.results()
    assert n_heavy > 20 and sa.heavy_units() == 0
    assert h1.tobytes() == h2.tobytes() and c1.tobytes() == c2.tobytes()
    sa.close()
    gref.close()


@pytest.fixture(scope="module")
def heavy_genome(tmp_path_factory):
    """a third of this genome is microsatellite: 3-letter buckets of 10^5 entries, multi-window / multi-task scans"""
    g = td.make_genome(seed=5, chr_lens=(2_000_000, 500_000), gc=0.45, microsats=3000, repeats=200, n_runs=4)
    fa = str(tmp_path_factory.mktemp("heavy") / "g.fa")
    td.write_fasta(fa, g)
    return g, fa


@pytest.mark.parametrize("pe", [False, True], ids=["se", "pe"])
def test_heavy_pipeline_large_buckets(pe, heavy_genome, oracle, extra=None, work_counters=True):
    g, fa = heavy_genome
    kw = dict(s=16, v=6, I=4, S=1, r=1)
    if pe:
        kw.update(m=28, x=500, pairend=1)
    kw.update(extra or {})
    oref = oracle.OracleRef(oracle.make_params(**kw), fasta_path=fa)
    gref = B.RefSeq(B.make_params(**kw)).Run_ConvertBinseq(fasta_path=fa).CreateIndex()
    assert np.diff(gref.index()[0].astype(np.int64)).max() > 50_000
    if not pe:
        reads = td.make_se_reads(g, 3000, 144, seed=8, sub_rate=0.03)
        sb, so = oracle.pack_reads([r["seq"] for r in reads])
        ores, ocnt = oracle.se_batch(oref, sb, so, threads=8)
        sa = B.SingleAlign(gref, len(reads)).set_work_counters(work_counters)
        sa.ImportBatchReads((sb, so)).Do_Batch()
        hits, cc = sa.results()
        assert sa.heavy_units() > 500
        test_heavy_pipeline_large_buckets.last_redo = sa.redo_units()
        assert np.array_equal(ores["n_hit"][:, :7], cc["n_hit"][:, :7]) and np.array_equal(ores["n_chit"][:, :7], cc["n_chit"][:, :7])
        has = ores["n_best"] > 0
        for f in ("chr", "loc", "best_class"):
            assert np.array_equal(ores[f][has], hits[f][has]), f
        assert not work_counters or _counters_off() or [int(x) for x in sa.counters()[:4]] == ocnt
        sa.close()
    else:
        pairs = td.make_pe_reads(g, 3000, 144, seed=8, sub_rate=0.01)
        s1, o1 = oracle.pack_reads([p["seq1"] for p in pairs])
        s2, o2 = oracle.pack_reads([p["seq2"] for p in pairs])
        ores, ocnt = oracle.pe_batch(oref, s1, o1, s2, o2, threads=8)
        pa = B.PairAlign(gref, len(pairs)).set_work_counters(work_counters)
        pa.ImportBatchReads((s1, o1), (s2, o2)).Do_Batch()
        out, ca, cb, npairs = pa.results()
        assert pa.heavy_units() > 500
        test_heavy_pipeline_large_buckets.last_redo = pa.redo_units()
        assert np.array_equal(ores["paired"], out["paired"]) and np.array_equal(ores["n_pairs"][:, :13], npairs[:, :13])
        for m, cnts in (("a", ca), ("b", cb)):
            assert np.array_equal(ores[m]["n_hit"][:, :7], cnts["n_hit"][:, :7]) and np.array_equal(ores[m]["n_chit"][:, :7], cnts["n_chit"][:, :7]), m
        pr = (ores["tmp"] == 0) & (ores["paired"] > 0)
        for f in ("a_chr", "a_loc", "b_chr", "b_loc", "insert", "na", "nb", "chain"):
            assert np.array_equal(ores["pick"][f][pr], out[f][pr]), f
        assert not work_counters or _counters_off() or [int(x) for x in pa.counters()[:4]] == ocnt
        test_heavy_pipeline_large_buckets.last_group_share = float(pa.counters()[15]) / max(1.0, float(pa.counters()[7]))
        pa.close()
    gref.close()
    oref.free()


@pytest.mark.parametrize("pe", [False, True], ids=["se", "pe"])
def test_heavy_units_redone_when_their_duplicate_set_overflows(pe, heavy_genome, oracle, monkeypatch):
    """the slabs of deferred units carry a small duplicate-suppression set; a unit that outgrows it (single-end RRBS in practice,
    forced here with a 64-key set) is handed back to the main kernel, which redoes it undeferred — records and work counters as
    if nothing had happened"""
    monkeypatch.setenv("BSX_HEAVY_KCAP", "64")
    test_heavy_pipeline_large_buckets(pe, heavy_genome, oracle)
    assert test_heavy_pipeline_large_buckets.last_redo > 20


@pytest.mark.parametrize("env", [dict(BSX_TAIL_TASKS="100000000", BSX_TAIL_GRID="64"), dict(BSX_TAIL_TASKS="0"), dict(BSX_HEAVY_GROUPS="2", BSX_TAIL_TASKS="100000000"),
                                 dict(BSX_SAME="0"), dict(BSX_SAME="0", BSX_SPREAD="1"), dict(BSX_SAME="2", BSX_SAME_GRID_DIV="64")],
                         ids=["tail_from_the_start_tiny_grid", "no_tail_mode", "two_groups_tail", "one_task_scan_kernel", "one_task_kernel_spread_order", "group_kernel_small_grid"])
@pytest.mark.parametrize("pe", [False, True], ids=["se", "pe"])
def test_heavy_pipeline_scan_grids_and_streams_do_not_matter(pe, env, heavy_genome, oracle, monkeypatch):
    """the scan kernels take their tasks in a grid-stride sweep, so any grid is correct: the tail mode (small grids on the group's
    high-priority stream) forced on from the first pass with a grid for 64 tasks, switched off, and combined with two unit groups —
    records and work counters equal the oracle's each time"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    test_heavy_pipeline_large_buckets(pe, heavy_genome, oracle)


@pytest.mark.parametrize("env", [dict(), dict(BSX_SAME="0")], ids=["group_scan_kernel", "one_task_scan_kernel"])
@pytest.mark.parametrize("pe", [False, True], ids=["se", "pe"])
def test_heavy_pipeline_without_work_counters(pe, env, heavy_genome, oracle, monkeypatch):
    """bsx_batch_set_work_counters(0) — the scan kernels without the early-out classification, what the command line runs: every record
    equals the oracle's (the work counters are not compared: they are what is switched off)"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    test_heavy_pipeline_large_buckets(pe, heavy_genome, oracle, work_counters=False)


def test_heavy_pipeline_runs_its_scan_in_groups(heavy_genome, oracle, monkeypatch):
    """the default WGBS scan kernel (k_hscan_same) finds tasks of equal window and read offset on the microsatellite genome and evaluates
    them as groups (counter 15: the share of the scan's candidates that ran in groups of two and more); with BSX_SAME=0 none do"""
    test_heavy_pipeline_large_buckets(True, heavy_genome, oracle)
    assert test_heavy_pipeline_large_buckets.last_group_share > 0.2, test_heavy_pipeline_large_buckets.last_group_share
    monkeypatch.setenv("BSX_SAME", "0")
    test_heavy_pipeline_large_buckets(True, heavy_genome, oracle)
    assert test_heavy_pipeline_large_buckets.last_group_share == 0.0


@pytest.mark.parametrize("extra", [dict(w=20), dict(w=3, r=0), dict(w=150, n=1), dict(r=0, v=3)], ids=["w20", "w3_r0", "w150_n1", "r0_v3"])
def test_heavy_pipeline_caps_and_early_returns(extra, heavy_genome, oracle):
    """-w caps and -r 0 on reads with hundreds of hits: the 64-at-a-time survivor acceptance has to cut its groups at
    the first survivor that lowers the threshold or ends the call, and re-publish the rest of the window"""
    test_heavy_pipeline_large_buckets(False, heavy_genome, oracle, extra)
    if "n" not in extra:
        test_heavy_pipeline_large_buckets(True, heavy_genome, oracle, extra)


HEAVY_SEEDS = list(range(1, 9 + EXTRA_FUZZ))   # (BSX_EXTRA_FUZZ=N: N more option draws, for a one-off soak)


@pytest.mark.parametrize("seed", HEAVY_SEEDS)
def test_heavy_pipeline_random_options(seed, heavy_genome, oracle):
    """seeded option draws on the microsatellite genome: windows, restarts, grouped acceptance and caps under other
    thresholds, hit caps, strand modes, read-length caps and insert ranges"""
    import random
    rng = random.Random(1000 + seed)
    extra = dict(v=rng.choice([3, 4, 5, 6, 8]), w=rng.choice([1, 5, 40, 300, 1000]), r=rng.choice([0, 1, 1]), S=rng.randint(1, 50), L=rng.choice([144, 144, 110]))
    pe = seed % 2 == 0
    if pe:
        extra.update(m=rng.choice([0, 28, 150]), x=rng.choice([300, 500, 800]))
    else:
        extra.update(n=rng.choice([0, 1]))
    test_heavy_pipeline_large_buckets(pe, heavy_genome, oracle, extra)


def test_heavy_pipeline_degenerate_genome(oracle, tmp_path):
    """worst case for the scan: a genome that is almost one microsatellite with a poly-T and a poly-A arm — every read hits
    buckets of 10^5..10^6 entries on both strands, thousands of equal hits, caps and threshold lowering all the time"""
    import random
    rng = random.Random(3)
    def mutate(s, rate):
        return "".join(c if rng.random() > rate else rng.choice("ACGT") for c in s)
    chr1 = mutate("TG" * 300_000, 0.01) + "".join(rng.choice("ACGT") for _ in range(20_000)) + mutate("T" * 150_000, 0.02)
    chr2 = mutate("CA" * 200_000, 0.01) + mutate("A" * 100_000, 0.02) + mutate("TTG" * 60_000, 0.01)
    g = [("chr1", chr1), ("chr2", chr2)]
    fa = str(tmp_path / "g.fa")
    td.write_fasta(fa, g)
    for pe, extra in ((False, dict(n=1)), (True, dict(m=28, x=500, pairend=1)), (False, dict(w=30, r=0))):
        kw = dict(s=16, v=6, I=4, S=1, r=1)
        kw.update(extra)
        oref = oracle.OracleRef(oracle.make_params(**kw), fasta_path=fa)
        gref = B.RefSeq(B.make_params(**kw)).Run_ConvertBinseq(fasta_path=fa).CreateIndex()
        assert np.diff(gref.index()[0].astype(np.int64)).max() > 100_000
        if not pe:
            reads = td.make_se_reads(g, 400, 144, seed=8, sub_rate=0.02, strands=("++", "-+", "+-", "--"))
            sb, so = oracle.pack_reads([r["seq"] for r in reads])
            ores, ocnt = oracle.se_batch(oref, sb, so, threads=8)
            sa = B.SingleAlign(gref, len(reads))
            sa.ImportBatchReads((sb, so)).Do_Batch()
            hits, cc = sa.results()
            assert sa.heavy_units() > 100
            assert np.array_equal(ores["n_hit"][:, :7], cc["n_hit"][:, :7]) and np.array_equal(ores["n_chit"][:, :7], cc["n_chit"][:, :7])
            has = ores["n_best"] > 0
            for f in ("chr", "loc", "best_class"):
                assert np.array_equal(ores[f][has], hits[f][has]), f
            assert np.array_equal(np.maximum(ores["n_best"], 0), hits["n_best"])
            assert _counters_off() or [int(x) for x in sa.counters()[:4]] == ocnt
            sa.close()
        else:
            pairs = td.make_pe_reads(g, 300, 144, seed=9, sub_rate=0.02)
            s1, o1 = oracle.pack_reads([p["seq1"] for p in pairs])
            s2, o2 = oracle.pack_reads([p["seq2"] for p in pairs])
            ores, ocnt = oracle.pe_batch(oref, s1, o1, s2, o2, threads=8)
            pa = B.PairAlign(gref, len(pairs))
            pa.ImportBatchReads((s1, o1), (s2, o2)).Do_Batch()
            out, ca, cb, npairs = pa.results()
            assert pa.heavy_units() > 80
            assert np.array_equal(ores["paired"], out["paired"]) and np.array_equal(ores["n_pairs"][:, :13], npairs[:, :13])
            for m, cnts in (("a", ca), ("b", cb)):
                assert np.array_equal(ores[m]["n_hit"][:, :7], cnts["n_hit"][:, :7]) and np.array_equal(ores[m]["n_chit"][:, :7], cnts["n_chit"][:, :7]), m
            pr = (ores["tmp"] == 0) & (ores["paired"] > 0)
            for f in ("a_chr", "a_loc", "b_chr", "b_loc", "insert", "na", "nb", "chain"):
                assert np.array_equal(ores["pick"][f][pr], out[f][pr]), f
            assert _counters_off() or [int(x) for x in pa.counters()[:4]] == ocnt
            pa.close()
        gref.close()
        oref.free()


def test_batch_reuse_and_argument_checks(edge_genome, oracle):
    """one device batch used for batches of different sizes, and the error codes of calls made out of order"""
    g, fa = edge_genome
    kw = dict(s=16, v=4, I=4, S=1, r=1)
    gref = B.RefSeq(B.make_params(**kw)).Run_ConvertBinseq(fasta_path=fa).CreateIndex()
    oref = oracle.OracleRef(oracle.make_params(**kw), fasta_path=fa)
    L = B.lib()
    sa = B.SingleAlign(gref, 500)
    assert L.bsx_batch_run(sa.h) != 0                                           # nothing uploaded yet
    assert L.bsx_batch_results_se(sa.h, np.zeros(4, B.HIT_DTYPE).ctypes.data, None) != 0  # nothing has run yet
    reads = td.make_se_reads(g, 500, 100, seed=21)
    al = oracle.OracleAligner(oref, leak_mode=0)
    for n in (500, 37, 1, 260):
        seqs = [r["seq"] for r in reads[:n]]
        sa.ImportBatchReads(seqs).Do_Batch()
        h, cc = sa.results()
        assert len(h) == n
        for i in (0, n // 2, n - 1):
            o = al.se(i, seqs[i])
            _cmp_read(o, h[i], cc[i], kw, i)
    with pytest.raises(Exception):
        sa.ImportBatchReads([r["seq"] for r in reads] + ["ACGT"])             # more units than the batch was created for
    with pytest.raises(Exception):
        sa.run_range(400, 200, sync=True)                                       # range past the uploaded units
    with pytest.raises(Exception):
        L2 = B.PairAlign(gref, 10)
        try:
            B._check(L.bsx_batch_upload_se(L2.h, 1, b"ACGT", np.array([0, 4], np.uint64).ctypes.data, None, 0))  # paired batch, single-end upload
        finally:
            L2.close()
    al.free(); sa.close(); gref.close(); oref.free()


def test_heavy_pipeline_small_pools(heavy_genome, oracle):
    """tiny task pool and tiny per-round capacity: requests get refused and retried, deferred units take several rounds"""
    B.lib().bsx_set_heavy_limits(97, 24)
    try:
        test_heavy_pipeline_large_buckets(True, heavy_genome, oracle)
        test_heavy_pipeline_large_buckets(False, heavy_genome, oracle)
    finally:
        B.lib().bsx_set_heavy_limits(32768, 524288)


@pytest.mark.parametrize("seed", [1, 5, 12])
def test_rrbs_heavy_pipeline_without_work_counters(seed, oracle, tmp_path_factory, monkeypatch):
    """RRBS lists through the scan kernels with the work counters off (BSX_WORK_COUNTERS=0 = bsx_batch_set_work_counters(0) for every batch):
    records as the oracle's"""
    monkeypatch.setenv("BSX_WORK_COUNTERS", "0")
    test_rrbs_through_the_heavy_pipeline(seed, oracle, tmp_path_factory)


@pytest.mark.parametrize("same", ["1", "2"], ids=["shared_scan_kernel", "group_scan_kernel"])
def test_rrbs_survivor_overflow_without_work_counters(same, oracle, tmp_path, monkeypatch):
    monkeypatch.setenv("BSX_WORK_COUNTERS", "0")
    monkeypatch.setenv("BSX_SAME", same)
    test_rrbs_shared_scan_runs_and_survivor_overflow(oracle, tmp_path)


@pytest.mark.parametrize("seed", [1, 5, 12])
def test_rrbs_through_the_group_scan_kernel(seed, oracle, tmp_path_factory, monkeypatch):
    """BSX_SAME=2: RRBS lists through k_hscan_same (groups formed by the pre-pass, the tag filter part of a group's signature,
    chromosome-local positions and per-entry strand copies in its loader) instead of k_hscan_shared — same records, same counters"""
    monkeypatch.setenv("BSX_SAME", "2")
    test_rrbs_through_the_heavy_pipeline(seed, oracle, tmp_path_factory)


def test_rrbs_group_scan_runs_and_survivor_overflow(oracle, tmp_path, monkeypatch):
    """the tandem families of the test below through k_hscan_same: survivor overflow per task, redone by the control kernel"""
    monkeypatch.setenv("BSX_SAME", "2")
    test_rrbs_shared_scan_runs_and_survivor_overflow(oracle, tmp_path)


def test_rrbs_shared_scan_runs_and_survivor_overflow(oracle, tmp_path):
    """k_hscan_shared: hundreds of RRBS reads that walk the same window of one bucket (two tandem families whose every copy carries
    the digestion site: 1500 and 300 near-identical copies), evaluated 16 per wave.  The 1500-copy family yields more survivors
    per scan task than the record holds (512): those tasks come back flagged and the control kernel redoes them with its own
    scan — counts, picks and the work counters must still equal the oracle's."""
    rng = np.random.default_rng(77)

    def family(n_copies, unit_len, div):
        unit = td.random_seq(rng, unit_len, 0.5).copy()
        out = []
        for _ in range(n_copies):
            cp = unit.copy()
            mut = rng.random(unit_len) < div
            cp[mut] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(mut.sum()))]
            out.append(b"CCGG" + cp.tobytes())
        return b"".join(out)
    flank = lambda n: td.random_seq(rng, n, 0.5).tobytes()
    chr1 = flank(30_000) + family(1500, 100, 0.004) + b"CCGG" + flank(20_000) + family(300, 120, 0.01) + b"CCGG" + flank(30_000)
    g = [("chr1", chr1.decode()), ("chr2", td.make_genome(seed=3, chr_lens=(60_000,), cpg_sites=150)[0][1])]
    fa = str(tmp_path / "g.fa")
    td.write_fasta(fa, g)
    kw = dict(D="C-CGG", v=2, S=7, r=1, w=1000, m=40, x=220)
    oref = oracle.OracleRef(oracle.make_params(**kw), fasta_path=fa)
    gref = B.RefSeq(B.make_params(**kw)).Run_ConvertBinseq(fasta_path=fa).CreateIndex()
    reads = [r for r in td.make_rrbs_reads(g, 1200, 75, seed=5, sub_rate=0.004)]
    sb, so = oracle.pack_reads([r["seq"] for r in reads])
    ores, ocnt = oracle.se_batch(oref, sb, so, threads=8)
    assert int(ores["n_hit"].sum(axis=1).max()) >= 1000  # the -w cap is reached: far more than 512 survivors in one task
    B.lib().bsx_set_heavy_threshold(200)
    try:
        sa = B.SingleAlign(gref, len(reads))
        sa.ImportBatchReads((sb, so)).Do_Batch()
        hits, cc = sa.results()
        assert sa.heavy_units() > 400
        nclass = kw["v"] + 1
        ok = ores["filtered"] == 0
        assert np.array_equal(ores["n_hit"][ok][:, :nclass], cc["n_hit"][ok][:, :nclass]) and np.array_equal(ores["n_chit"][ok][:, :nclass], cc["n_chit"][ok][:, :nclass])
        has = ok & (ores["n_best"] > 0)
        for f in ("chr", "loc", "best_class"):
            assert np.array_equal(ores[f][has], hits[f][has]), f
        assert _counters_off() or [int(x) for x in sa.counters()[:4]] == ocnt
        sa.close()
    finally:
        B.lib().bsx_set_heavy_threshold(0)
    gref.close()
    oref.free()


def test_rrbs_capacity_limit_is_flagged_and_counted(oracle, tmp_path, monkeypatch):
    """BSX_F_LIMIT (include/bsx.h): a single-end RRBS read that remembers more than the set's capacity of distinct in-threshold coordinates (2^18; the
    reference's std::set is unbounded, align.cpp:274) is flagged and counted (counter 16), nothing is written past the slab, and every OTHER read of the
    batch still equals the oracle's.  Forced with a 64-key set (BSX_KCAP) on the tandem families of the test above, undeferred; with the real capacity
    the same batch has no flagged read."""
    rng = np.random.default_rng(78)
    unit = td.random_seq(rng, 100, 0.5)
    fam = []
    for _ in range(600):
        cp = unit.copy()
        mut = rng.random(100) < 0.004
        cp[mut] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(mut.sum()))]
        fam.append(b"CCGG" + cp.tobytes())
    flank = lambda n: td.random_seq(rng, n, 0.5).tobytes()
    g = [("chr1", (flank(20_000) + b"".join(fam) + b"CCGG" + flank(20_000)).decode()), ("chr2", td.make_genome(seed=4, chr_lens=(60_000,), cpg_sites=150)[0][1])]
    fa = str(tmp_path / "g.fa")
    td.write_fasta(fa, g)
    kw = dict(D="C-CGG", v=2, S=7, r=1, w=1000, m=40, x=220)
    oref = oracle.OracleRef(oracle.make_params(**kw), fasta_path=fa)
    reads = td.make_rrbs_reads(g, 600, 75, seed=6, sub_rate=0.004)
    sb, so = oracle.pack_reads([r["seq"] for r in reads])
    ores, ocnt = oracle.se_batch(oref, sb, so, threads=8)
    B.lib().bsx_set_heavy_threshold(1 << 30)   # the main kernel's own set is the one with the capacity flag
    try:
        res = {}
        for cap in ("64", None):
            if cap:
                monkeypatch.setenv("BSX_KCAP", cap)
            else:
                monkeypatch.delenv("BSX_KCAP")
            gref = B.RefSeq(B.make_params(**kw)).Run_ConvertBinseq(fasta_path=fa).CreateIndex()
            sa = B.SingleAlign(gref, len(reads))
            sa.ImportBatchReads((sb, so)).Do_Batch()
            hits, cc = sa.results()
            res[cap] = (hits.copy(), cc.copy(), sa.counters().copy())
            sa.close()
            gref.close()
    finally:
        B.lib().bsx_set_heavy_threshold(0)
    hits, cc, cnt = res["64"]
    flagged = (hits["flags"] & 4) != 0
    assert 50 < int(flagged.sum()) < len(reads) and int(cnt[16]) == int(flagged.sum())
    ok = (ores["filtered"] == 0) & ~flagged
    nclass = kw["v"] + 1
    assert np.array_equal(ores["n_hit"][ok][:, :nclass], cc["n_hit"][ok][:, :nclass]) and np.array_equal(ores["n_chit"][ok][:, :nclass], cc["n_chit"][ok][:, :nclass])
    has = ok & (ores["n_best"] > 0)
    for f in ("chr", "loc", "best_class"):
        assert np.array_equal(ores[f][has], hits[f][has]), f
    hits, cc, cnt = res[None]   # the shipped capacity: nothing flagged, everything equal
    assert int(cnt[16]) == 0 and not ((hits["flags"] & 4) != 0).any()
    ok = ores["filtered"] == 0
    assert np.array_equal(ores["n_hit"][ok][:, :nclass], cc["n_hit"][ok][:, :nclass])
    has = ok & (ores["n_best"] > 0)
    for f in ("chr", "loc", "best_class"):
        assert np.array_equal(ores[f][has], hits[f][has]), f
    assert _counters_off() or [int(x) for x in cnt[:4]] == ocnt
    oref.free()


# ---- the same comparisons with the work counters off: the main kernel then takes its context prefilter (wave_scan_range<.., CTX>: a candidate is compared with
# the 32 reference nt left and right of its seed — words that come with the index entry — before it gathers anything) and the scan kernels skip the early-out
# classification.  Every hit list, pair list, class count and pick must be what the oracle says; only the work counters are not compared.
def test_golden_sets_without_work_counters(case, monkeypatch):
    monkeypatch.setenv("BSX_WORK_COUNTERS", "0")
    test_alignment_vs_oracle_and_golden(case)


@pytest.mark.parametrize("name,kw,spec", EDGE, ids=[e[0] for e in EDGE])
def test_edge_cases_without_work_counters(name, kw, spec, edge_genome, oracle, monkeypatch):
    monkeypatch.setenv("BSX_WORK_COUNTERS", "0")
    test_edge_cases_vs_oracle(name, kw, spec, edge_genome, oracle)


@pytest.mark.parametrize("seed", list(range(101, 161 + EXTRA_FUZZ)))
def test_random_option_combinations_without_work_counters(seed, edge_genome, oracle, monkeypatch):
    monkeypatch.setenv("BSX_WORK_COUNTERS", "0")
    test_random_option_combinations_vs_oracle(seed, edge_genome, oracle)


def test_empty_and_degenerate_inputs_without_work_counters(edge_genome, oracle, monkeypatch):
    monkeypatch.setenv("BSX_WORK_COUNTERS", "0")
    test_empty_and_degenerate_inputs(edge_genome, oracle)


# ---- the heavy pipeline's replay with the work counters off (round 6): a lowered threshold (align.cpp:278) no longer ends a window — the survivors behind the
# event are replayed under the new threshold in the same control pass (snp_align_heavy, BSX_EVENT_CONTINUE).  The cap / early-return cases, the option draws and
# the degenerate genome are where thresholds fall all the time.
@pytest.mark.parametrize("extra", [dict(w=20), dict(w=3, r=0), dict(w=150, n=1), dict(r=0, v=3), dict(w=1), dict(w=2, v=8)], ids=["w20", "w3_r0", "w150_n1", "r0_v3", "w1", "w2_v8"])
def test_heavy_pipeline_caps_and_early_returns_without_work_counters(extra, heavy_genome, oracle, monkeypatch):
    monkeypatch.setenv("BSX_WORK_COUNTERS", "0")
    test_heavy_pipeline_large_buckets(False, heavy_genome, oracle, extra, work_counters=False)
    if "n" not in extra:
        test_heavy_pipeline_large_buckets(True, heavy_genome, oracle, extra, work_counters=False)


@pytest.mark.parametrize("seed", HEAVY_SEEDS)
def test_heavy_pipeline_random_options_without_work_counters(seed, heavy_genome, oracle, monkeypatch):
    monkeypatch.setenv("BSX_WORK_COUNTERS", "0")
    test_heavy_pipeline_random_options(seed, heavy_genome, oracle)


def test_heavy_pipeline_degenerate_genome_without_work_counters(oracle, tmp_path, monkeypatch):
    monkeypatch.setenv("BSX_WORK_COUNTERS", "0")
    test_heavy_pipeline_degenerate_genome(oracle, tmp_path)


def test_context_table_is_optional_and_droppable(heavy_genome, oracle):
    """bsx_ref_set_context / bsx_ref_context_bytes / bsx_ref_drop_context: the table of the main kernel's prefilter is built by the headroom rule,
    never when switched off, is refused a drop while a batch lives, and records are the same with and without it"""
    g, fa = heavy_genome
    kw = dict(s=16, v=6, I=4, S=1, r=1, m=28, x=500, pairend=1)
    pairs = td.make_pe_reads(g, 1500, 144, seed=18, sub_rate=0.01)
    s1, o1 = oracle.pack_reads([p["seq1"] for p in pairs])
    s2, o2 = oracle.pack_reads([p["seq2"] for p in pairs])
    outs = []
    for context, headroom, expect in ((0, 0, False), (1, 1 << 50, False), (1, 0, True), (2, 0, True)):
        gref = B.RefSeq(B.make_params(**kw)).Run_ConvertBinseq(fasta_path=fa).CreateIndex(context=context, headroom=headroom)
        assert (gref.context_bytes > 0) == expect, (context, headroom, gref.context_bytes)
        if expect:
            assert gref.context_bytes >= 16 * gref.n_entries
        pa = B.PairAlign(gref, len(pairs)).set_work_counters(False)
        pa.ImportBatchReads((s1, o1), (s2, o2)).Do_Batch()
        out, ca, cb, npairs = pa.results()
        outs.append((out.tobytes(), ca.tobytes(), cb.tobytes(), npairs.tobytes()))
        if expect:
            with pytest.raises(Exception):
                gref.drop_context()   # a batch of the reference exists
        pa.close()
        gref.drop_context()
        assert gref.context_bytes == 0
        pa = B.PairAlign(gref, len(pairs)).set_work_counters(False)   # the same reference without its table: the plain main kernel
        pa.ImportBatchReads((s1, o1), (s2, o2)).Do_Batch()
        out, ca, cb, npairs = pa.results()
        assert (out.tobytes(), ca.tobytes(), cb.tobytes(), npairs.tobytes()) == outs[-1]
        pa.close()
        gref.close()
    assert all(o == outs[0] for o in outs)
