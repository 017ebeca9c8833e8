"""The plain-C oracle (oracle/bsx_oracle.c) against the golden vectors recorded from the real reference.
CPU only.  This is what pins the oracle on machines where /root/reference does not exist."""
import ctypes as C

import numpy as np
import pytest

import golden_util as G


def _tuples(lst):
    return [tuple(x) for x in lst]


@pytest.fixture(scope="module", params=G.CONFIGS)
def case(request, oracle):
    meta, arr, fasta = G.load(request.param)
    p = oracle.make_params(**meta["kw"])
    oref = oracle.OracleRef(p, fasta_path=fasta)
    yield meta, arr, oref, oracle
    oref.free()


def test_packed_reference(case):
    meta, arr, oref, O = case
    assert np.array_equal(oref.refcat()[400:-400], arr["refcat"])
    assert np.array_equal(oref.crefcat()[400:-400], arr["crefcat"])
    assert not oref.refcat()[:400].any() and not oref.refcat()[-400:].any()
    assert np.array_equal(oref.anchor(), arr["anchor"])
    assert np.array_equal(oref.chr_size(), arr["chr_size"])
    assert np.array_equal(oref.rc_offset(), arr["rc_offset"])
    assert np.array_equal(oref.blocks(), arr["blocks"])


def test_seed_index(case):
    meta, arr, oref, O = case
    if "D" in meta["kw"]:
        keys, n, _ = G.sparse_index(oref.bucket_off())
        assert np.array_equal(keys, arr["idx_keys"]) and np.array_equal(n, arr["idx_n"])
        assert np.array_equal(oref.rrbs_entries(), arr["idx_entries"])
        assert np.array_equal(oref.sites(0), arr["sites0"]) and np.array_equal(oref.sites(1), arr["sites1"])
    else:
        keys, n, nf = G.sparse_index(oref.bucket_off(), oref.bucket_nfwd())
        assert np.array_equal(keys, arr["idx_keys"]) and np.array_equal(n, arr["idx_n"]) and np.array_equal(nf, arr["idx_nfwd"])
        assert np.array_equal(oref.entries(), arr["idx_entries"])


def check_state(exp, got, nclass, tag):
    assert exp["filtered"] == got.filtered and exp["len"] == got.len and exp["raw_len"] == got.raw_len, tag
    if exp["filtered"]:
        return
    assert exp["read_max_snp_num"] == got.read_max_snp_num and exp["seedseg_num"] == got.seedseg_num, tag
    assert exp["snp_thres"] == got.snp_thres, tag
    n = exp["seedseg_num"]
    if exp["flag_chain"]:
        assert exp["seed_start_array"][:n] == list(got.seed_start_array)[:n], tag
        assert exp["seedindex"][:n] == list(got.seedindex)[:n] and exp["seedcount"][:n] == list(got.seedcount)[:n], tag
    if exp["cflag_chain"]:
        assert exp["cseed_start_array"][:n] == list(got.cseed_start_array)[:n], tag
        assert exp["cseedindex"][:n] == list(got.cseedindex)[:n] and exp["cseedcount"][:n] == list(got.cseedcount)[:n], tag
    assert exp["n_hit"][:nclass] == list(got.n_hit)[:nclass] and exp["n_chit"][:nclass] == list(got.n_chit)[:nclass], tag


def test_alignment(case):
    """per read: planner state, every hit list, every pair list, and the hit the reference's formatter printed"""
    meta, arr, oref, O = case
    kw = meta["kw"]
    nclass = kw["v"] + 1
    names = oref.names()
    al = O.OracleAligner(oref, leak_mode=1)  # golden records were produced by one reference object in file order
    for i, (r, e) in enumerate(zip(meta["reads"], meta["expected"])):
        if meta["kind"] == "se":
            got = al.se(i, r["seq"], r["qual"])
            check_state(e, got, nclass, (meta["config"], i))
            if e["filtered"]:
                continue
            for w in range(nclass):
                for o in (0, 1):
                    assert _tuples(e["hits"][w][o]) == al.se_hits(o, w, len(e["hits"][w][o])), (i, w, o)
            f = e["line"].split("\t")
            if e["line"] and f[2] != "*":
                assert f[2] == names[got.chr >> 1] and int(f[3]) == got.loc + 1 and f[11] == "NM:i:%d" % got.best_class, (i, e["line"])
                assert "ZS:Z:" + "+-"[got.chr & 1] + "+-"[got.chain] in e["line"]
            else:
                assert got.n_best == 0 or (kw["r"] == 0 and got.n_best > 1)
        else:
            got = al.pe(i, r["seq1"], r["seq2"], r["qual1"], r["qual2"])
            assert (e["paired"], e["tmp"], e["n_pairs"]) == (got.paired, got.tmp, list(got.n_pairs)), i
            check_state(e["a"], got.a, nclass, (i, "a"))
            check_state(e["b"], got.b, nclass, (i, "b"))
            for mate in (0, 1):
                ee = e["ab"[mate]]
                if ee["filtered"]:
                    continue
                for w in range(nclass):
                    for o in (0, 1):
                        assert _tuples(ee["hits"][w][o]) == al.pe_hits(mate, o, w, len(ee["hits"][w][o])), (i, mate, w, o)
            for w, pl in enumerate(e["pairs"]):
                assert _tuples(pl) == al.pe_pairs(w, len(pl)), (i, w)
            lines = [x.split("\t") for x in e["line"].split("\n") if x]
            if e["paired"] and e["tmp"] == 0:
                pk = got.pick
                f = lines[0]
                if pk.insert >= got.a.len and pk.insert >= got.b.len:  # no read-through trimming (pairs.cpp:296-306)
                    assert f[2] == names[pk.a.chr >> 1] and int(f[3]) == pk.a.loc + 1 and int(f[7]) == pk.b.loc + 1, (i, f)
                    assert abs(int(f[8])) == pk.insert and f[11] == "NM:i:%d" % pk.na
            else:
                for f in lines:
                    m = got.a if int(f[1]) & 0x40 else got.b
                    if f[2] != "*":
                        assert f[2] == names[m.chr >> 1] and int(f[3]) == m.loc + 1 and f[11] == "NM:i:%d" % m.best_class, (i, f)
    al.free()


def test_bsmap_binary_output(case):
    """the SAM file written by the real `bsmap -p 1`: every mapped SE line must be the oracle's pick"""
    meta, arr, oref, O = case
    if meta["kind"] != "se":
        pytest.skip("PE text is covered by test_alignment via the harness lines")
    names = oref.names()
    al = O.OracleAligner(oref, leak_mode=1)
    by_name = {}
    for ln in meta["bsmap_sam"].split("\n"):
        if ln and not ln.startswith("@"):
            f = ln.split("\t")
            by_name[f[0]] = f
    n_checked = 0
    for i, r in enumerate(meta["reads"]):
        got = al.se(i, r["seq"], r["qual"])
        f = by_name.get(r["name"])
        mapped = (not got.filtered) and (got.n_best == 1 or (got.n_best > 1 and meta["kw"]["r"] == 1))
        assert (f is not None) == mapped, (i, r["name"])
        if f is not None:
            assert f[2] == names[got.chr >> 1] and int(f[3]) == got.loc + 1 and f[11] == "NM:i:%d" % got.best_class
            n_checked += 1
    assert n_checked > 100
    al.free()


def test_myrand_known_answers(oracle):
    """utilities.cpp:44-48 (the -S != 0 branch) restated in exact integer arithmetic"""
    p = oracle.make_params(S=1)
    s = C.c_uint32(1)
    vals = [oracle.lib().bso_myrand(C.byref(p), i, C.byref(s)) for i in (0, 1, 2, 12345)]

    def ref(i, seed=1):
        M = (1 << 64) - 1
        v = ((i + seed * 1000000) * 3935559000370003845 + 2691343689449507681) & M
        v ^= v >> 21
        v ^= (v << 37) & M
        v ^= v >> 4
        v = (v * 4768777513237032717) & M
        v ^= (v << 20) & M
        v ^= v >> 41
        v ^= (v << 5) & M
        return v & 0xffffffff
    assert vals == [ref(i) for i in (0, 1, 2, 12345)]


def test_batch_threads_equal_serial(oracle):
    meta, arr, fasta = G.load("c2_se100_n1")
    p = oracle.make_params(**meta["kw"])
    oref = oracle.OracleRef(p, fasta_path=fasta)
    seqs = [r["seq"][:144] for r in meta["reads"]]
    buf, off = oracle.pack_reads(seqs)
    r1, c1 = oracle.se_batch(oref, buf, off, threads=1)
    r4, c4 = oracle.se_batch(oref, buf, off, threads=4)
    assert r1.tobytes() == r4.tobytes() and c1 == c4
    oref.free()
