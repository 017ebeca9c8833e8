"""BASELINE-sized checks (hg38-sized synthetic genome, 3.09 Gbp, 1.48 G index entries; the bench workload).
Parity at this size:
  * the seed index itself: the oracle packs and indexes the genome TEXT pulled back from the device (its restatement of
    dbseq.cpp:308-481, one thread, minutes) and every packed word, bucket offset, forward count and entry of the GPU-built
    index (k_enumerate -> radix sort -> k_boundaries, 1.476 G entries) must equal it — once per session for WGBS (-s 16 -I 4,
    all WGBS configs share it), and for the RRBS {tag, loc} index of C4;
  * WHOLE batches (131 072 units of C2 / C3, 65 536 of C2 -n 1 / C4 / C5) re-aligned by the oracle's batch driver on the host
    cores against the ORACLE-BUILT reference + index: every record field and the four work counters; C5 also in exact mode
    against the oracle's `-p 1` state; a summary goes to gpurun_out/validate/r04_validate_<cfg>.json (copied into profiles/);
  * idempotence: the same batch twice gives byte-identical records and counters;
  * partition invariance: aligning in two halves equals aligning the whole batch;
  * path invariance: results do not depend on which units go through the heavy pipeline (default threshold vs none);
  * geometry: reads were sampled as 50..480 nt fragments; reported pairs are same-chromosome, insert in range, >95 % unique;
  * counter closure: units processed == units submitted."""
import hashlib
import time

import numpy as np
import pytest

import bsmap_amd as B

pytestmark = pytest.mark.gpu

HG38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
        135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
        46709983, 50818468, 156040895, 57227415]
KW = dict(s=16, v=6, I=4, m=28, x=500, S=1, r=1, pairend=1)
N = 131072


@pytest.fixture(scope="module")
def big():
    ref = B.RefSeq(B.make_params(**KW)).synthetic(HG38, seed=38).CreateIndex()
    pa = B.PairAlign(ref, N)
    pa.synth_reads(N, 144, seed=11)
    pa.Do_Batch()
    out, ca, cb, npairs = pa.results()
    cnt = pa.counters().copy()
    HEAVY["c3"] = int(pa.heavy_units())   # of THIS run (later tests run parts of the batch again)
    yield ref, pa, out.copy(), ca.copy(), cb.copy(), npairs.copy(), cnt
    pa.close()
    ref.close()


HEAVY = {}


def _genome_text(ref):
    parts = []
    for c, nm in enumerate(ref.names()):
        parts += [np.frombuffer(f">{nm}\n".encode(), np.uint8), ref.synth_bytes(c), np.frombuffer(b"\n", np.uint8)]
    return np.concatenate(parts).tobytes()


class _OracleWgbs:
    """The oracle's OWN reference + WGBS index of the hg38-sized genome, built once per session from the genome text (-s 16 -I 4:
    what every WGBS config here uses), and the proof that a GPU-built reference equals it."""

    def __init__(self, oracle):
        self.oracle, self.base, self.info = oracle, None, None

    def _build(self, ref):
        text = _genome_text(ref)
        t0 = time.time()
        self.base = self.oracle.OracleRef(self.oracle.make_params(**KW), fasta_text=text)
        self.build_s = time.time() - t0

    def check(self, ref, name):
        """every word / offset / count / entry of the GPU-built reference `ref` against the oracle's build"""
        if self.base is None:
            self._build(ref)
        o = self.base
        f, c = ref.words()
        assert np.array_equal(f, o.refcat()) and np.array_equal(c, o.crefcat())
        a, s, r = ref.info()
        assert np.array_equal(a, o.anchor()) and np.array_equal(s, o.chr_size()) and np.array_equal(r, o.rc_offset())
        del f, c
        off, nf, ent = ref.index()
        assert len(ent) == o.r.n_entries == ref.n_entries
        assert np.array_equal(off, o.bucket_off()), "bucket offsets"
        assert np.array_equal(nf, o.bucket_nfwd()), "forward counts"
        assert np.array_equal(ent, o.entries()), "entries"
        if self.info is None:
            sizes = np.diff(off.astype(np.int64))
            self.info = dict(entries=int(len(ent)), buckets=int(len(nf)), non_empty_buckets=int((sizes > 0).sum()), largest_bucket=int(sizes.max()),
                             entries_sha256=hashlib.sha256(ent.tobytes()).hexdigest(), bucket_off_sha256=hashlib.sha256(off.tobytes()).hexdigest(),
                             oracle_build_s=round(self.build_s, 1), checked=[])
        self.info["checked"].append(name)
        import wholebatch as W
        W.record("index", self.info)

    def ref_for(self, kw):
        """the oracle-built arrays under another config's parameters (same -s / -I)"""
        o = self.base
        return self.oracle.OracleRef.wrap(self.oracle.make_params(**kw), o.refcat(), o.crefcat(), o.anchor(), o.chr_size(), o.rc_offset(),
                                          o.bucket_off(), o.bucket_nfwd(), o.entries())


@pytest.fixture(scope="session")
def oracle_wgbs(oracle):
    w = _OracleWgbs(oracle)
    yield w
    if w.base is not None:
        w.base.free()


def test_gpu_built_index_equals_the_oracles_own_build(big, oracle_wgbs):
    """RefSeq::CreateIndex at BASELINE size (dbseq.cpp:308-481): 1.476 G entries through 64-bit enumeration, 32-bit keys and a
    radix sort on the device against the oracle's count / prefix / fill from the genome text"""
    ref = big[0]
    oracle_wgbs.check(ref, "c3")
    assert oracle_wgbs.info["entries"] > 1_400_000_000


def test_sizes_and_closure(big):
    ref, pa, out, ca, cb, npairs, cnt = big
    assert ref.n_entries > 1_400_000_000 and ref.n_words > 190_000_000
    assert int(cnt[4]) == N and pa.heavy_units() > N // 200
    assert (out["unpaired_out"] == 0).mean() > 0.99


def test_idempotent_and_partition_invariant(big):
    ref, pa, out, ca, cb, npairs, cnt = big
    pa.reset_counters()
    pa.Do_Batch()
    out2, ca2, cb2, np2 = pa.results()
    assert out2.tobytes() == out.tobytes() and ca2.tobytes() == ca.tobytes() and np2.tobytes() == npairs.tobytes()
    assert np.array_equal(pa.counters()[:7], cnt[:7])
    pa.run_range(N // 2, N // 2, sync=True)
    pa.run_range(0, N // 2, sync=True)
    out3, ca3, cb3, np3 = pa.results()
    assert out3.tobytes() == out.tobytes() and cb3.tobytes() == cb.tobytes()


def test_records_do_not_depend_on_the_work_counters(big):
    """with the work counters off (bsx_batch_set_work_counters: the scan kernels skip the early-out classification, the control kernel its
    count-only walks — what the command line and bench.py's timed region run) every record is byte-identical and the aligned counts agree"""
    ref, pa, out, ca, cb, npairs, cnt = big
    pa.set_work_counters(False)
    try:
        pa.reset_counters()
        pa.Do_Batch()
        out2, ca2, cb2, np2 = pa.results()
        c2 = pa.counters()
    finally:
        pa.set_work_counters(True)
    assert out2.tobytes() == out.tobytes() and ca2.tobytes() == ca.tobytes() and cb2.tobytes() == cb.tobytes() and np2.tobytes() == npairs.tobytes()
    assert np.array_equal(c2[4:7], cnt[4:7]) and int(c2[2]) < int(cnt[2])   # (sum_w is what is no longer counted)
    assert pa.heavy_units() == HEAVY["c3"]


def test_heavy_path_invariance(big):
    ref, pa, out, ca, cb, npairs, cnt = big
    # 4096 units through the one-wave path only (threshold so high that nothing is deferred)
    B.lib().bsx_set_heavy_threshold(1 << 30)
    try:
        pa.reset_counters()
        pa.run_range(0, 2048, sync=True)
        assert pa.heavy_units() == 0
        o2, a2, b2, n2 = pa.results()
    finally:
        B.lib().bsx_set_heavy_threshold(0)
    assert o2[:2048].tobytes() == out[:2048].tobytes() and a2[:2048].tobytes() == ca[:2048].tobytes() and n2[:2048].tobytes() == npairs[:2048].tobytes()


def test_whole_batch_equals_oracle_c3(big, oracle, oracle_wgbs):
    """every unit of the batch re-aligned by the oracle's batch driver against the ORACLE's own reference + index (shown equal
    to the device's in the test above): every record field and the four work counters (the roofline numerator)"""
    import wholebatch as W
    ref, pa, out, ca, cb, npairs, cnt = big
    if oracle_wgbs.base is None:
        oracle_wgbs.check(ref, "c3")
    oref = oracle_wgbs.ref_for(KW)
    ores, ocnt, t_cpu = W.run_oracle(oracle, oref, pa, True, False, N)
    bad, info = W.compare_pe(ores, out, ca, cb, npairs, KW["v"] + 1)
    W.record("c3", dict(info, units=N, oracle_s=round(t_cpu, 1), counters_gpu=[int(x) for x in cnt[:4]], counters_oracle=ocnt, mismatching_fields=bad,
                        options=KW, heavy_units=HEAVY["c3"], reference="oracle-built from the genome text"))
    assert not bad, bad
    assert [int(x) for x in cnt[:4]] == ocnt


def test_blocks_of_a_bench_sized_batch_equal_the_oracle(big, oracle, oracle_wgbs):
    """bench.py's own step — 2^22 pairs, read seed 3, unit ids from 0, the pools the bench gives it, work counters off as in the timed region — run as
    ONE batch (its deferred units are grouped as in the bench: 37 K of them), and 32 blocks of 4 096 consecutive units spread over the batch (2^17 units)
    re-aligned by the oracle against its own reference + index: every record field.  (Round 4 validated the 2^20-pair bench batch with a tool outside
    the suite, profiles/r04h_validate_full_c3.json.)"""
    import wholebatch as W
    ref = big[0]
    if oracle_wgbs.base is None:
        oracle_wgbs.check(ref, "c3")
    oref = oracle_wgbs.ref_for(KW)
    NB, BLK, NBLK = 1 << 22, 4096, 32
    L = B.lib()
    u, t = B.default_heavy_limits(B.make_params(**KW), NB, True)
    L.bsx_set_heavy_limits(u, t)
    try:
        pa = B.PairAlign(ref, NB).set_work_counters(False)
    finally:
        L.bsx_set_heavy_limits(0, 0)
    try:
        pa.synth_reads(NB, 144, seed=3)
        t0 = time.time()
        pa.Do_Batch()
        t_gpu = time.time() - t0
        out, ca, cb, npairs = pa.results()
        heavy = int(pa.heavy_units())
        hlist = np.sort(pa.heavy_list().astype(np.int64))
        b1, o1 = pa.download_reads(0)
        b2, o2 = pa.download_reads(1)
    finally:
        pa.close()
    bad_all, t_cpu, placed = {}, 0.0, 0
    # the deferred units carry 97 % of the step's candidates: the comparison has to hold a good number of them — those that fall into the blocks, and 256 more
    # taken evenly from the deferred list, each re-aligned alone (the pick RNG is a function of the unit's own index)
    spans = [(k * (NB // NBLK) + 17 * k, k * (NB // NBLK) + 17 * k + BLK) for k in range(NBLK)]   # (not on any power-of-two grid)
    in_blocks = int(sum(np.searchsorted(hlist, hi) - np.searchsorted(hlist, lo) for lo, hi in spans))
    extra = [int(u) for u in hlist[:: max(1, len(hlist) // 256)][:256] if not any(lo <= u < hi for lo, hi in spans)]
    assert len(hlist) == heavy and in_blocks >= 1000 and len(extra) >= 200, (len(hlist), heavy, in_blocks, len(extra))
    for lo, hi in spans + [(u, u + 1) for u in extra]:
        t0 = time.time()
        ores, _ = oracle.pe_batch(oref, b1[int(o1[lo]):int(o1[hi])], (o1[lo:hi + 1] - o1[lo]).copy(), b2[int(o2[lo]):int(o2[hi])], (o2[lo:hi + 1] - o2[lo]).copy(),
                                  first_index=lo, threads=W.usable_cpus())
        t_cpu += time.time() - t0
        bad, info = W.compare_pe(ores, out[lo:hi], ca[lo:hi], cb[lo:hi], npairs[lo:hi], KW["v"] + 1)
        placed += info["paired_out"]
        for f, n in bad.items():
            bad_all[f] = bad_all.get(f, 0) + n
    W.record("bench_step_c3", dict(units_in_batch=NB, units_compared=BLK * NBLK + len(extra), blocks=NBLK, heavy_units=heavy, deferred_units_compared=in_blocks + len(extra), paired_out=placed, oracle_s=round(t_cpu, 1), do_batch_s=round(t_gpu, 3),
                                   mismatching_fields=bad_all, options=KW, work_counters=False, reference="oracle-built from the genome text"))
    assert heavy > NB // 200 and placed > 0.95 * BLK * NBLK
    assert not bad_all, bad_all


def test_pair_geometry(big):
    """the sampler draws fragments of 50..480 nt from one chromosome: reported pairs must respect that geometry"""
    ref, pa, out, ca, cb, npairs, cnt = big
    pr = out["unpaired_out"] == 0
    # both mates on the same chromosome copy, insert within [50, 480] as sampled, mates facing each other
    assert np.array_equal(out["a_chr"][pr], out["b_chr"][pr])
    ins = out["insert"][pr]
    assert ins.min() >= 28 and ins.max() <= 500 and 250 < np.median(ins) < 350
    uniq = pr & (out["n_pairs"] == 1)
    assert uniq.mean() > 0.95


def test_two_batches_in_flight(big):
    """two device batches driven by two host threads at the same time (what bench.py and the command-line driver do):
    same records as the batch that ran alone"""
    import threading
    ref, pa, out, ca, cb, npairs, cnt = big
    others = [B.PairAlign(ref, N) for _ in range(2)]
    res = [None, None]
    try:
        for o in others:
            o.synth_reads(N, 144, seed=11)

        def work(j):
            for _ in range(2):
                others[j].Do_Batch()
            res[j] = tuple(x.copy() for x in others[j].results())

        th = [threading.Thread(target=work, args=(j,)) for j in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for j in range(2):
            o2, a2, b2, n2 = res[j]
            assert o2.tobytes() == out.tobytes() and a2.tobytes() == ca.tobytes() and b2.tobytes() == cb.tobytes() and n2.tobytes() == npairs.tobytes()
    finally:
        for o in others:
            o.close()


# ---- the other BASELINE configurations at the same scale --------------------------------------------------------------
ADAPTER = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"
OTHER = {
    # C2: 1x100 WGBS, -v 4; with -n 1 all four strands are searched from one read
    "c2": dict(kw=dict(s=16, v=4, I=4, S=1, r=1), pe=False, L=100, kind=0, n=131072),
    "c2_n1": dict(kw=dict(s=16, v=4, I=4, S=1, r=1, n=1), pe=False, L=100, kind=0, n=65536),
    # C5: 2x144 with 3' adapters and low-quality tails, -A <adapter> -q 20 (variable read lengths after trimming)
    "c5": dict(kw=dict(s=16, v=6, I=4, m=28, x=500, S=1, r=1, pairend=1, q=20, A=[ADAPTER]), pe=True, L=144, kind=1, n=65536),
    # C4: RRBS -D C-CGG (seed 12, interval 1 forced), 1x75 reads at digestion sites; site tables + {tag, loc} index of the whole genome
    "c4": dict(kw=dict(D="C-CGG", S=1, r=1), pe=False, L=75, kind=2, n=65536),
}


@pytest.fixture(scope="module", params=sorted(OTHER))
def other(request):
    cfg = OTHER[request.param]
    ref = B.RefSeq(B.make_params(**cfg["kw"])).synthetic(HG38, seed=38).CreateIndex()
    al = (B.PairAlign if cfg["pe"] else B.SingleAlign)(ref, cfg["n"])
    al.synth_reads(cfg["n"], cfg["L"], seed=17, kind=cfg["kind"])
    al.Do_Batch()
    res = tuple(x.copy() for x in al.results())
    cnt = al.counters().copy()
    HEAVY[request.param] = (int(al.heavy_units()), int(al.redo_units()))   # of THIS run
    yield request.param, cfg, ref, al, res, cnt
    al.close()
    ref.close()


def test_other_configs_closure_idempotence_partition(other):
    name, cfg, ref, al, res, cnt = other
    n = cfg["n"]
    assert int(cnt[4]) == n
    assert ref.n_words > 190_000_000 and ref.n_entries > (1_000_000 if name == "c4" else 1_400_000_000)
    al.reset_counters()
    al.Do_Batch()
    res2 = al.results()
    for a, b in zip(res, res2):
        assert a.tobytes() == b.tobytes()
    assert np.array_equal(al.counters()[:7], cnt[:7])
    al.run_range(n // 2, n // 2, sync=True)
    al.run_range(0, n // 2, sync=True)
    for a, b in zip(res, al.results()):
        assert a.tobytes() == b.tobytes()
    al.set_work_counters(False)   # (what the command line runs: records must not depend on the counters)
    try:
        al.reset_counters()
        al.Do_Batch()
        for a, b in zip(res, al.results()):
            assert a.tobytes() == b.tobytes()
        assert np.array_equal(al.counters()[4:7], cnt[4:7])
    finally:
        al.set_work_counters(True)
    if cfg["pe"]:
        out = res[0]
        placed = (out["unpaired_out"] == 0).mean()
        assert placed > 0.9, placed            # trimmed pairs still pair up
        assert len(np.unique(out["a"]["len"])) > 20   # trimming really produced a spread of read lengths
    else:
        hits = res[0]
        assert (hits["n_best"] > 0).mean() > (0.85 if name != "c4" else 0.8)


def _oracle_ref(name, cfg, ref, oracle, oracle_wgbs):
    kw = cfg["kw"]
    if name == "c4":   # RRBS: the oracle's OWN packing, site tables and index of the genome text
        text = _genome_text(ref)
        oref = oracle.OracleRef(oracle.make_params(**kw), fasta_text=text)
        del text
        assert sum(len(oref.sites(c)) for c in range(ref.n_chr)) == sum(len(ref.sites(c)) for c in range(ref.n_chr))
        assert np.array_equal(oref.sites(3), ref.sites(3))
        assert np.array_equal(oref.rrbs_entries(), ref.index()[2])
        return oref
    oracle_wgbs.check(ref, name)   # this config's own GPU-built reference + index against the oracle's build
    return oracle_wgbs.ref_for(kw)


def test_other_configs_whole_batch_equals_oracle(other, oracle, oracle_wgbs):
    """every unit of the batch re-aligned by the oracle's batch driver against the oracle's OWN build of reference + index from
    the 3.1 GB genome text (WGBS and RRBS alike; the device's copy is compared with it first) — every record field and the four
    work counters; C5 also in exact mode (bsx_batch_set_leak_exact) against the oracle's `-p 1` state (leak_mode 1)"""
    import wholebatch as W
    name, cfg, ref, al, res, cnt = other
    kw, n = cfg["kw"], cfg["n"]
    nclass = kw.get("v", 2) + 1
    quals = cfg["kind"] == 1
    oref = _oracle_ref(name, cfg, ref, oracle, oracle_wgbs)
    try:
        ores, ocnt, t_cpu = W.run_oracle(oracle, oref, al, cfg["pe"], quals, n)
        if cfg["pe"]:
            bad, info = W.compare_pe(ores, res[0], res[1], res[2], res[3], nclass)
        else:
            bad, info = W.compare_se(ores, res[0], res[1], nclass)
        rec = dict(info, units=n, oracle_s=round(t_cpu, 1), counters_gpu=[int(x) for x in cnt[:4]], counters_oracle=ocnt, mismatching_fields=bad,
                   options={k: v for k, v in kw.items()}, heavy_units=HEAVY[name][0], redo_units=HEAVY[name][1],
                   reference="oracle-built from the genome text")
        if name == "c5":   # the only BASELINE config whose reads leak planner state ((len - I + 1) % S == 0 after trimming)
            ex = B.PairAlign(ref, n).set_leak_exact()
            try:
                ex.synth_reads(n, cfg["L"], seed=17, kind=cfg["kind"])
                ex.Do_Batch()
                eres = ex.results()
                ecnt = [int(x) for x in ex.counters()[:4]]
            finally:
                ex.close()
            lres, lcnt, t_leak = W.run_oracle(oracle, oref, al, True, True, n, leak_mode=1)
            ebad, _ = W.compare_pe(lres, eres[0], eres[1], eres[2], eres[3], nclass)
            S, I = kw["s"], kw["I"]
            leaky = sum(int((((lres[m]["len"] - I + 1) % S == 0) & (lres[m]["filtered"] == 0)).sum()) for m in ("a", "b"))
            differ = int((eres[0].tobytes() != res[0].tobytes()))
            rec["exact_mode"] = dict(oracle_s=round(t_leak, 1), counters_gpu=ecnt, counters_oracle=lcnt, mismatching_fields=ebad, leaky_reads=leaky,
                                     records_differ_from_default_mode=bool(differ))
            W.record(name, rec)
            assert leaky > n // 50
            assert not ebad, ebad
            assert ecnt == lcnt
        else:
            W.record(name, rec)
        assert not bad, bad
        assert [int(x) for x in cnt[:4]] == ocnt
    finally:
        oref.free()
