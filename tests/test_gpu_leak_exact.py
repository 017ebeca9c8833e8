"""-p 1 exact mode (bsx_batch_set_leak_exact): reads with (len - I + 1) % S == 0 are planned with the state earlier reads of
their stream left in the reference's never-reset SingleAlign members (align.h:82-91).  With the mode on, the HIP path must
equal the REAL reference's single-threaded results for EVERY read of the golden sets — nothing excluded — and the oracle in
call order (leak_mode 1), including planner arrays, every hit / pair list and the work counters; a batch split in two with
the first half attached as history must give the same records as the whole batch."""
import os

import numpy as np
import pytest

import bsmap_amd as B
import bsx_testdata as td
import golden_util as G

pytestmark = pytest.mark.gpu


def _n_leaky(meta):
    kw = meta["kw"]
    if "D" in kw:
        return 0
    S, I = kw.get("s", 16), kw.get("I", 4)
    f = lambda e: (not e["filtered"]) and (e["len"] - I + 1) % S == 0
    if meta["kind"] == "se":
        return sum(f(e) for e in meta["expected"])
    return sum(f(e["a"]) or f(e["b"]) for e in meta["expected"])


def _counters_off():
    """BSX_WORK_COUNTERS=0: batches are created with the work counters off — the exact mode then runs the main kernel WITH the context prefilter
    (k_align<PE, EXACT, CTX>, round 6); records are compared, the counters are not"""
    return os.environ.get("BSX_WORK_COUNTERS") == "0"


@pytest.mark.parametrize("name", G.CONFIGS)
def test_exact_mode_equals_reference_goldens_for_every_read(name, oracle):
    meta, arr, fasta = G.load(name)
    kw = meta["kw"]
    nclass = kw["v"] + 1
    reads = meta["reads"]
    oref = oracle.OracleRef(oracle.make_params(**kw), fasta_path=fasta)
    gref = B.RefSeq(B.make_params(**kw)).Run_ConvertBinseq(fasta_path=fasta).CreateIndex()
    al = oracle.OracleAligner(oref, leak_mode=1)
    n_checked = 0
    if meta["kind"] == "se":
        sa = B.SingleAlign(gref, len(reads), debug=True).set_leak_exact()
        sa.ImportBatchReads([r["seq"] for r in reads], [r["qual"] for r in reads]).Do_Batch()
        hits, cc = sa.results()
        for i, r in enumerate(reads):
            o = al.se(i, r["seq"], r["qual"])
            e = meta["expected"][i]
            assert bool(e["filtered"]) == bool(hits[i]["flags"] & 1), i
            if e["filtered"]:
                continue
            n = o.seedseg_num
            st, od = sa.debug_plan(i)
            if o.flag_chain:
                assert list(st[0][:n]) == list(o.seed_start_array)[:n] and list(od[0][:n]) == list(o.seedindex)[:n], i
            if o.cflag_chain:
                assert list(st[1][:n]) == list(o.cseed_start_array)[:n] and list(od[1][:n]) == list(o.cseedindex)[:n], i
            assert e["n_hit"][:nclass] == list(cc[i]["n_hit"][:nclass]) and e["n_chit"][:nclass] == list(cc[i]["n_chit"][:nclass]), i
            for w in range(nclass):
                for orient in (0, 1):
                    assert [tuple(x) for x in e["hits"][w][orient]] == sa.debug_hits(i, 0, orient, w), (i, w, orient)
            n_checked += 1
        assert _counters_off() or [int(x) for x in sa.counters()[:4]] == al.counters()
        sa.close()
    else:
        pa = B.PairAlign(gref, len(reads), debug=True).set_leak_exact()
        pa.ImportBatchReads([r["seq1"] for r in reads], [r["seq2"] for r in reads], [r["qual1"] for r in reads], [r["qual2"] for r in reads]).Do_Batch()
        out, ca, cb, npairs = pa.results()
        for i, r in enumerate(reads):
            o = al.pe(i, r["seq1"], r["seq2"], r["qual1"], r["qual2"])
            e = meta["expected"][i]
            g = out[i]
            assert (bool(e["a"]["filtered"]), bool(e["b"]["filtered"])) == (bool(g["a"]["flags"] & 1), bool(g["b"]["flags"] & 1)), i
            for mate, (om, ccm) in enumerate(((o.a, ca), (o.b, cb))):
                if om.filtered:
                    continue
                assert list(om.n_hit)[:nclass] == list(ccm[i]["n_hit"][:nclass]) and list(om.n_chit)[:nclass] == list(ccm[i]["n_chit"][:nclass]), (i, mate)
                for w in range(nclass):
                    for orient in (0, 1):
                        nn = (om.n_chit if orient else om.n_hit)[w]
                        assert pa.debug_hits(i, mate, orient, w) == al.pe_hits(mate, orient, w, nn), (i, mate, w, orient)
            if not e["a"]["filtered"] and not e["b"]["filtered"]:
                assert e["paired"] == g["paired"] and e["n_pairs"][:2 * nclass - 1] == list(npairs[i][:2 * nclass - 1]), i
                for w, pl in enumerate(e["pairs"]):
                    assert [tuple(x) for x in pl] == pa.debug_pairs(i, w), (i, w)
                n_checked += 1
        assert _counters_off() or [int(x) for x in pa.counters()[:4]] == al.counters()
        pa.close()
    al.free(); gref.close(); oref.free()
    assert n_checked > len(reads) // 2
    if name.startswith(("c5", "c2")):
        assert _n_leaky(meta) > 0   # the sets the round-1 comparisons had to thin out


@pytest.mark.parametrize("pe", [False, True])
def test_exact_mode_history_and_variable_lengths(pe, oracle):
    """variable read lengths (1/16 of them leak), -n 1 so that both orientations carry state; the second half of the input as
    its own batch with the first half attached as history equals the whole input in one batch and the oracle in call order"""
    g = td.make_genome(seed=21, chr_lens=(150_000, 60_000), gc=0.45)
    fasta = td.fasta_text(g)
    kw = dict(s=16, v=5, I=4, S=2, r=1, n=1) if not pe else dict(s=16, v=5, I=4, S=2, r=1, m=28, x=500, pairend=1)
    oref = oracle.OracleRef(oracle.make_params(**kw), fasta_text=fasta)
    gref = B.RefSeq(B.make_params(**kw)).Run_ConvertBinseq(fasta_text=fasta).CreateIndex()
    rng = np.random.default_rng(5)
    n = 3000
    if pe:
        pairs = td.make_pe_reads(g, n, 144, seed=9)
        la, lb = rng.integers(40, 145, n), rng.integers(40, 145, n)
        s1 = [p["seq1"][:int(l)] for p, l in zip(pairs, la)]
        s2 = [p["seq2"][:int(l)] for p, l in zip(pairs, lb)]
        al = oracle.OracleAligner(oref, leak_mode=1)
        exp = [al.pe(i, a, b) for i, (a, b) in enumerate(zip(s1, s2))]
        whole = B.PairAlign(gref, n).set_leak_exact()
        whole.ImportBatchReads(s1, s2).Do_Batch()
        out, ca, cb, npairs = whole.results()
        assert _counters_off() or [int(x) for x in whole.counters()[:4]] == al.counters()
        for i, o in enumerate(exp):
            assert o.paired == out[i]["paired"] and list(o.n_pairs)[:11] == list(npairs[i][:11]), i
            assert list(o.a.n_hit)[:6] == list(ca[i]["n_hit"][:6]) and list(o.b.n_chit)[:6] == list(cb[i]["n_chit"][:6]), i
        h = n // 2 + 7
        half = B.PairAlign(gref, n).set_leak_exact()
        half.set_history(s1[h - 64:h], None, s2[h - 64:h], None)
        half.ImportBatchReads(s1[h:], s2[h:], first_index=h).Do_Batch()
        o2, a2, b2, n2 = half.results()
        assert o2.tobytes() == out[h:].tobytes() and a2.tobytes() == ca[h:].tobytes() and b2.tobytes() == cb[h:].tobytes() and n2.tobytes() == npairs[h:].tobytes()
        # and the default (zero-state) mode differs somewhere on such input: the mode is doing something
        plain = B.PairAlign(gref, n)
        plain.ImportBatchReads(s1, s2).Do_Batch()
        assert plain.results()[1].tobytes() != ca.tobytes() or plain.results()[2].tobytes() != cb.tobytes()
        for b in (whole, half, plain):
            b.close()
    else:
        reads = td.make_se_reads(g, n, 144, seed=9)
        ln = rng.integers(30, 145, n)
        ss = [r["seq"][:int(l)] for r, l in zip(reads, ln)]
        al = oracle.OracleAligner(oref, leak_mode=1)
        exp = [al.se(i, s) for i, s in enumerate(ss)]
        whole = B.SingleAlign(gref, n).set_leak_exact()
        whole.ImportBatchReads(ss).Do_Batch()
        hits, cc = whole.results()
        assert _counters_off() or [int(x) for x in whole.counters()[:4]] == al.counters()
        for i, o in enumerate(exp):
            assert list(o.n_hit)[:6] == list(cc[i]["n_hit"][:6]) and list(o.n_chit)[:6] == list(cc[i]["n_chit"][:6]), i
            if o.n_best > 0:
                assert (o.chr, o.loc, o.best_class) == (hits[i]["chr"], hits[i]["loc"], hits[i]["best_class"]), i
        h = n // 2 + 7
        half = B.SingleAlign(gref, n).set_leak_exact()
        half.set_history(ss[h - 64:h])
        half.ImportBatchReads(ss[h:], first_index=h).Do_Batch()
        h2, c2 = half.results()
        assert h2.tobytes() == hits[h:].tobytes() and c2.tobytes() == cc[h:].tobytes()
        plain = B.SingleAlign(gref, n)
        plain.ImportBatchReads(ss).Do_Batch()
        assert plain.results()[1].tobytes() != cc.tobytes()
        for b in (whole, half, plain):
            b.close()
    al.free(); gref.close(); oref.free()


@pytest.mark.parametrize("pe", [False, True])
def test_exact_mode_state_chained_from_batch_to_batch(pe, oracle):
    """an input cut into four batches of unequal size, each started from the state the one before it returns
    (bsx_batch_get_leak_state -> bsx_batch_set_leak_state), equals the whole input in one batch and the oracle in call order — with a
    setter that lies thousands of reads back (one long read, then only reads that never set the offset)"""
    g = td.make_genome(seed=23, chr_lens=(120_000,), gc=0.45)
    fasta = td.fasta_text(g)
    kw = dict(s=16, v=4, I=4, S=2, r=1, n=1) if not pe else dict(s=16, v=6, I=4, S=2, r=1, m=28, x=500, pairend=1)
    oref = oracle.OracleRef(oracle.make_params(**kw), fasta_text=fasta)
    gref = B.RefSeq(B.make_params(**kw)).Run_ConvertBinseq(fasta_text=fasta).CreateIndex()
    rng = np.random.default_rng(8)
    n = 6000
    leaky = [131, 115, 99, 83, 67, 51]
    la = [144, 143, 142, 141, 140] + [int(x) for x in rng.choice(leaky, n - 5)]
    lb = [140, 144, 139, 138, 137] + [int(x) for x in rng.choice(leaky, n - 5)]
    la[4000], lb[2500] = 100, 126   # one more setter per stream late in the input (the offsets they leave differ)
    cuts = [0, 1700, 1701, 4300, n]

    def offset_setters(seqs, readset):
        """full-length reads whose planned start offsets are all > 0 (so the offset they leave behind is not the initial 0): on a small
        random genome most reads tie at offset 0 (first minimum wins, align.cpp:462), and a state of 0 would test nothing"""
        al = oracle.OracleAligner(oref, 0)
        good = []
        for i, sq in enumerate(seqs[:1500]):
            if len(sq) < 144:
                continue
            o = al.se(i, sq, readset=readset)
            st = list(o.cseed_start_array if readset == 2 else o.seed_start_array)[:max(1, o.seedseg_num)]
            if not o.filtered and min(st) > 0:
                good.append(sq)
        al.free()
        return good

    if pe:
        pairs = td.make_pe_reads(g, n, 144, seed=12)
        g1, g2 = offset_setters([p["seq1"] for p in pairs], 1), offset_setters([p["seq2"] for p in pairs], 2)
        assert g1 and g2
        for k, pos in enumerate((4, 4000)):
            pairs[pos]["seq1"], la[pos] = g1[k % len(g1)], 144
        for k, pos in enumerate((4, 2500)):
            pairs[pos]["seq2"], lb[pos] = g2[k % len(g2)], 144
        s1 = [p["seq1"][:l] for p, l in zip(pairs, la)]
        s2 = [p["seq2"][:l] for p, l in zip(pairs, lb)]
        b1, o1 = oracle.pack_reads(s1)
        b2, o2 = oracle.pack_reads(s2)
        exp, ecnt = oracle.pe_batch(oref, b1, o1, b2, o2, threads=4, leak_mode=1)
        whole = B.PairAlign(gref, n).set_leak_exact()
        whole.ImportBatchReads(s1, s2).Do_Batch()
        out, ca, cb, npairs = whole.results()
        assert _counters_off() or [int(x) for x in whole.counters()[:4]] == ecnt
        assert np.array_equal(exp["paired"], out["paired"]) and np.array_equal(exp["a"]["n_hit"][:, :7], ca["n_hit"][:, :7]) and np.array_equal(exp["b"]["n_chit"][:, :7], cb["n_chit"][:, :7])
        st = None
        part = B.PairAlign(gref, n).set_leak_exact()
        for lo, hi in zip(cuts, cuts[1:]):
            part.set_leak_state(st)
            part.ImportBatchReads(s1[lo:hi], s2[lo:hi], first_index=lo).Do_Batch()
            o_, a_, b_, n_ = part.results()
            assert o_.tobytes() == out[lo:hi].tobytes() and a_.tobytes() == ca[lo:hi].tobytes() and b_.tobytes() == cb[lo:hi].tobytes(), lo
            st = part.get_leak_state()
        assert st.tobytes() == whole.get_leak_state().tobytes() and st.any()
        plain = B.PairAlign(gref, n)
        plain.ImportBatchReads(s1, s2).Do_Batch()
        assert _counters_off() or [int(x) for x in plain.counters()[:3]] != ecnt[:3]   # the plans differ (the hits they lead to need not)
        for b in (whole, part, plain):
            b.close()
    else:
        reads = td.make_se_reads(g, n, 144, seed=12, junk_frac=0.0)
        good = offset_setters([r["seq"] for r in reads], 0)
        assert good
        for k, pos in enumerate((4, 4000)):
            reads[pos]["seq"], la[pos] = good[k % len(good)], 144
        ss = [r["seq"][:l] for r, l in zip(reads, la)]
        b1, o1 = oracle.pack_reads(ss)
        exp, ecnt = oracle.se_batch(oref, b1, o1, threads=4, leak_mode=1)
        whole = B.SingleAlign(gref, n).set_leak_exact()
        whole.ImportBatchReads(ss).Do_Batch()
        hits, cc = whole.results()
        assert _counters_off() or [int(x) for x in whole.counters()[:4]] == ecnt
        assert np.array_equal(exp["n_hit"][:, :5], cc["n_hit"][:, :5]) and np.array_equal(exp["n_chit"][:, :5], cc["n_chit"][:, :5])
        st = None
        part = B.SingleAlign(gref, n).set_leak_exact()
        for lo, hi in zip(cuts, cuts[1:]):
            part.set_leak_state(st)
            part.ImportBatchReads(ss[lo:hi], first_index=lo).Do_Batch()
            h_, c_ = part.results()
            assert h_.tobytes() == hits[lo:hi].tobytes() and c_.tobytes() == cc[lo:hi].tobytes(), lo
            st = part.get_leak_state()
        assert st.tobytes() == whole.get_leak_state().tobytes() and st.any()
        plain = B.SingleAlign(gref, n)
        plain.ImportBatchReads(ss).Do_Batch()
        assert _counters_off() or [int(x) for x in plain.counters()[:3]] != ecnt[:3]   # the plans differ (the hits they lead to need not)
        for b in (whole, part, plain):
            b.close()
    gref.close(); oref.free()


def test_exact_mode_when_no_read_ever_sets_the_offset():
    """uniform 51-nt reads with -s 16 -I 4: (len - I + 1) % S == 0 for every read, nothing ever sets the start offset and nothing writes
    the tail entries — the search for a setter must give up through the block summaries, not walk the whole stream for every read
    (2^18 reads: a per-read walk is 10^10 steps); the state stays the initial one, so the records equal the default mode's"""
    import time
    g = td.make_genome(seed=24, chr_lens=(200_000,), gc=0.5)
    kw = dict(s=16, v=2, I=4, S=1, r=1)
    gref = B.RefSeq(B.make_params(**kw)).Run_ConvertBinseq(fasta_text=td.fasta_text(g)).CreateIndex()
    n = 1 << 18
    base = td.make_se_reads(g, 4096, 51, seed=3, junk_frac=0.0)
    ss = [base[i % 4096]["seq"] for i in range(n)]
    ex = B.SingleAlign(gref, n).set_leak_exact()
    ex.ImportBatchReads(ss)
    ex.Do_Batch()
    t0 = time.time()
    ex.Do_Batch()
    dt = time.time() - t0
    h1, c1 = ex.results()
    pl = B.SingleAlign(gref, n)
    pl.ImportBatchReads(ss).Do_Batch()
    h2, c2 = pl.results()
    assert h1.tobytes() == h2.tobytes() and c1.tobytes() == c2.tobytes()
    st = ex.get_leak_state().view(np.uint32)
    assert not st[2 * 2 * 160:].any()                      # no start offset was ever set
    key00 = st[:160]                                       # mate stream 0, forward orientation: the last read's hashes, nothing behind them
    assert key00[:51 - 16 + 1].any() and not key00[51 - 16 + 1:].any()
    assert dt < 5.0, dt
    ex.close(); pl.close(); gref.close()


# ---- the same with the work counters off: the exact mode on top of the context prefilter (what `BSX_P1_EXACT=1 bsmap` runs)
@pytest.mark.parametrize("name", G.CONFIGS)
def test_exact_mode_goldens_without_work_counters(name, oracle, monkeypatch):
    monkeypatch.setenv("BSX_WORK_COUNTERS", "0")
    test_exact_mode_equals_reference_goldens_for_every_read(name, oracle)


@pytest.mark.parametrize("pe", [False, True])
def test_exact_mode_history_without_work_counters(pe, oracle, monkeypatch):
    monkeypatch.setenv("BSX_WORK_COUNTERS", "0")
    test_exact_mode_history_and_variable_lengths(pe, oracle)


@pytest.mark.parametrize("pe", [False, True])
def test_exact_mode_chained_state_without_work_counters(pe, oracle, monkeypatch):
    monkeypatch.setenv("BSX_WORK_COUNTERS", "0")
    test_exact_mode_state_chained_from_batch_to_batch(pe, oracle)
