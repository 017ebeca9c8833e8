"""The oracle's batch driver in leak_mode 1 (`bsmap -p 1` planner state, align.h:82-91) works in parallel chunks that re-establish
the state by a planner-only replay (oracle/bsx_oracle.c, batch_job).  It must equal ONE aligner object fed read by read in input
order — the form that tests/test_oracle_vs_reference.py pins against the real reference.  CPU only."""
import numpy as np
import pytest

import bsx_testdata as td


def _mixed_lengths(rng, reads, key, lens):
    for r in reads:
        L = int(rng.choice(lens))
        for k in key:
            r[k] = r[k][:L]
    return reads


@pytest.mark.parametrize("threads", [1, 5])
def test_se_leak_batch_equals_sequential(oracle, threads):
    O = oracle
    g = td.make_genome(seed=21, chr_lens=(90_000, 30_000), gc=0.45)
    kw = dict(s=16, v=4, I=4, S=1, r=1, n=1)
    oref = O.OracleRef(O.make_params(**kw), fasta_text=td.fasta_text(g))
    rng = np.random.default_rng(5)
    # (len - 3) % 16 == 0 leaks: 99, 83, 67, 51; long and short non-leaky reads in between, a run of leaky ones at the start
    reads = td.make_se_reads(g, 3000, 100, seed=9)
    lens = [99, 83, 67, 51, 100, 100, 96, 90, 77, 60]
    for i, r in enumerate(reads):
        L = 99 if i < 40 else int(rng.choice(lens))
        r["seq"] = r["seq"][:L]
    buf, off = O.pack_reads([r["seq"] for r in reads])
    al = O.OracleAligner(oref, leak_mode=1)
    seq_res = [al.se(i, r["seq"]) for i, r in enumerate(reads)]
    seq_cnt = [al.counters()[k] // 1 for k in range(4)]
    res, cnt = O.se_batch(oref, buf, off, threads=threads, leak_mode=1)
    zero, _ = O.se_batch(oref, buf, off, threads=threads, leak_mode=0)
    n_diff = 0
    for i, s in enumerate(seq_res):
        b = res[i]
        assert (s.filtered, s.len, s.n_best, s.best_class, s.chr, s.loc) == (b["filtered"], b["len"], b["n_best"], b["best_class"], b["chr"], b["loc"]), i
        assert list(s.seed_start_array) == list(b["seed_start_array"]) and list(s.cseed_start_array) == list(b["cseed_start_array"]), i
        assert list(s.n_hit) == list(b["n_hit"]) and list(s.n_chit) == list(b["n_chit"]), i
        n_diff += list(s.seed_start_array) != list(zero[i]["seed_start_array"]) or list(s.cseed_start_array) != list(zero[i]["cseed_start_array"])
    # OracleAligner.counters() sums its two objects; only the first was used
    assert [int(x) for x in cnt] == [int(x) for x in seq_cnt]
    assert n_diff > 20  # the state really leaks in this input
    al.free(); oref.free()


def test_pe_leak_batch_equals_sequential(oracle):
    O = oracle
    g = td.make_genome(seed=22, chr_lens=(80_000,), gc=0.5)
    kw = dict(s=16, v=6, I=4, S=1, r=1, m=28, x=500, pairend=1)
    oref = O.OracleRef(O.make_params(**kw), fasta_text=td.fasta_text(g))
    rng = np.random.default_rng(6)
    pairs = td.make_pe_reads(g, 1500, 150, seed=10)
    lens = [131, 115, 99, 144, 140, 125, 101, 88]
    for p in pairs:
        p["seq1"] = p["seq1"][:int(rng.choice(lens))]
        p["seq2"] = p["seq2"][:int(rng.choice(lens))]
    s1, o1 = O.pack_reads([p["seq1"] for p in pairs])
    s2, o2 = O.pack_reads([p["seq2"] for p in pairs])
    al = O.OracleAligner(oref, leak_mode=1)
    seq_res = [al.pe(i, p["seq1"], p["seq2"]) for i, p in enumerate(pairs)]
    seq_cnt = al.counters()
    res, cnt = O.pe_batch(oref, s1, o1, s2, o2, threads=6, leak_mode=1)
    for i, s in enumerate(seq_res):
        b = res[i]
        assert s.paired == b["paired"] and list(s.n_pairs) == list(b["n_pairs"]), i
        for m in ("a", "b"):
            x, y = getattr(s, m), b[m]
            assert list(x.seed_start_array) == list(y["seed_start_array"]) and list(x.cseed_start_array) == list(y["cseed_start_array"]), (i, m)
            assert list(x.n_hit) == list(y["n_hit"]) and list(x.n_chit) == list(y["n_chit"]), (i, m)
        if s.paired and s.tmp == 0:
            assert (s.pick.a.chr, s.pick.a.loc, s.pick.b.chr, s.pick.b.loc, s.pick.insert) == tuple(int(b["pick"][f]) for f in ("a_chr", "a_loc", "b_chr", "b_loc", "insert")), i
    assert [int(x) for x in cnt] == [int(x) for x in seq_cnt]
    al.free(); oref.free()
