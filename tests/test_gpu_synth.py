"""The device-side synthetic workload (bench input) is held to the same parity bar: the oracle packs and indexes the
text of the synthetic genome and aligns the device-sampled reads; everything must match the GPU path bit for bit."""
import os

import numpy as np
import pytest

import bsmap_amd as B

pytestmark = pytest.mark.gpu

LENS = (600_000, 450_000, 120_000)


@pytest.fixture(scope="module")
def synth(oracle):
    out = {}
    for name, kw in (("se", dict(s=16, v=4, I=4, S=1, r=1)), ("pe", dict(s=16, v=6, I=4, S=1, r=1, m=28, x=500, pairend=1))):
        gref = B.RefSeq(B.make_params(**kw)).synthetic(LENS, seed=38).CreateIndex()
        text = "".join(f">{n}\n{gref.synth_text(c)}\n" for c, n in enumerate(gref.names()))
        oref = oracle.OracleRef(oracle.make_params(**kw), fasta_text=text)
        out[name] = (kw, gref, oref, text)
    yield out
    for kw, gref, oref, _ in out.values():
        gref.close()
        oref.free()


def test_synthetic_genome_matches_oracle_packer(synth):
    kw, gref, oref, text = synth["se"]
    a, s, r = gref.info()
    assert np.array_equal(a, oref.anchor()) and np.array_equal(s, oref.chr_size()) and np.array_equal(r, oref.rc_offset())
    f, c = gref.words()
    assert np.array_equal(f, oref.refcat()) and np.array_equal(c, oref.crefcat())
    assert np.array_equal(gref.blocks(), oref.blocks())
    off, nf, ent = gref.index()
    assert np.array_equal(off, oref.bucket_off()) and np.array_equal(nf, oref.bucket_nfwd()) and np.array_equal(ent, oref.entries())
    # composition sanity: ~5 % N, GC ~ 41-46 %, CpG depleted, skewed 3-letter buckets
    seq = text.split("\n")[1]
    n = seq.count("N") / len(seq)
    assert 0.03 < n < 0.08
    acgt = len(seq) - seq.count("N")
    gc = (seq.count("G") + seq.count("C")) / acgt
    assert 0.38 < gc < 0.50
    cpg = seq.count("CG") * acgt / max(1, seq.count("C") * seq.count("G"))
    assert cpg < 0.6
    assert np.diff(off.astype(np.int64)).max() > 200


def test_synth_se_reads_align_like_oracle(synth, oracle):
    kw, gref, oref, _ = synth["se"]
    sa = B.SingleAlign(gref, 30000)
    sa.synth_reads(30000, 100, seed=2)
    sa.run_range(0, 10000, sync=True)
    sa.run_range(10000, 20000, sync=True)
    hits, cc = sa.results()
    buf, off = sa.download_reads(0)
    ores, ocnt = oracle.se_batch(oref, buf, off, threads=4)
    assert np.array_equal(ores["n_hit"][:, :5], cc["n_hit"][:, :5]) and np.array_equal(ores["n_chit"][:, :5], cc["n_chit"][:, :5])
    has = ores["n_best"] > 0
    for f in ("chr", "loc", "best_class"):
        assert np.array_equal(ores[f][has], hits[f][has]), f
    assert os.environ.get("BSX_WORK_COUNTERS") == "0" or [int(x) for x in sa.counters()[:4]] == ocnt
    assert has.mean() > 0.9  # the sampler produces alignable bisulfite reads
    sa.close()


def test_synth_pe_reads_align_like_oracle(synth, oracle):
    kw, gref, oref, _ = synth["pe"]
    pa = B.PairAlign(gref, 20000)
    pa.synth_reads(20000, 144, seed=3)
    pa.Do_Batch()
    out, ca, cb, npairs = pa.results()
    b1, o1 = pa.download_reads(0)
    b2, o2 = pa.download_reads(1)
    ores, ocnt = oracle.pe_batch(oref, b1, o1, b2, o2, threads=4)
    assert np.array_equal(ores["paired"], out["paired"]) and np.array_equal(ores["n_pairs"][:, :13], npairs[:, :13])
    pr = (ores["tmp"] == 0) & (ores["paired"] > 0)
    for f in ("a_chr", "a_loc", "b_chr", "b_loc", "insert", "na", "nb", "chain"):
        assert np.array_equal(ores["pick"][f][pr], out[f][pr]), f
    assert os.environ.get("BSX_WORK_COUNTERS") == "0" or [int(x) for x in pa.counters()[:4]] == ocnt
    assert pr.mean() > 0.85
    # mates are proper pairs: a maps to ++/-+ and b to the opposite read orientation on the same reference copy
    assert np.array_equal(out["a_chr"][pr], out["b_chr"][pr])
    pa.close()
