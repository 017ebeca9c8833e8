#!/usr/bin/env python3
"""One-off wide parity check at the bench scale (GPU box): the hg38-sized synthetic genome, a 2^20-pair batch aligned on
the GPU, and the first K pairs (a contiguous block, so that the oracle's pthread batch driver can take them) re-aligned by
the oracle on the host cores against the same reference + index copied back from HBM.  Every record field is compared.
usage: validate_fullsize.py [--pairs 1048576] [--check 200000]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bsmap_amd as B
from oracle import oracle_ffi as O

HG38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
        135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
        46709983, 50818468, 156040895, 57227415]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1 << 20)
    ap.add_argument("--check", type=int, default=200000)
    ap.add_argument("--genome-seed", type=int, default=38)
    ap.add_argument("--read-seed", type=int, default=77)
    ap.add_argument("--se", action="store_true", help="single reads of 100 nt with the C2 options (-v 4) instead of pairs")
    ap.add_argument("--opts", default="", help="extra option letters as k=v,k=v (e.g. v=4,w=50,r=0)")
    a = ap.parse_args()
    kw = dict(s=16, v=6, I=4, m=28, x=500, S=1, r=1, pairend=1)
    if a.se:
        kw = dict(s=16, v=4, I=4, S=1, r=1)
    kw.update({k: int(v) for k, v in (kv.split("=") for kv in a.opts.split(",") if kv)})
    if a.se:
        return validate_se(a, kw)
    ref = B.RefSeq(B.make_params(**kw)).synthetic(HG38, seed=a.genome_seed).CreateIndex()
    pa = B.PairAlign(ref, a.pairs)
    pa.synth_reads(a.pairs, 144, seed=a.read_seed)
    pa.Do_Batch()
    out, ca, cb, npairs = pa.results()
    K = min(a.check, a.pairs)
    f, c = ref.words(); an, sz, rc = ref.info(); off, nf, ent = ref.index()
    oref = O.OracleRef.wrap(O.make_params(**kw), f, c, an, sz, rc, off, nf, ent)
    b1, o1 = pa.download_reads(0); b2, o2 = pa.download_reads(1)
    s1, s2 = b1[:int(o1[K])], b2[:int(o2[K])]
    t0 = time.time()
    ores, ocnt = O.pe_batch(oref, s1, o1[:K + 1].copy(), s2, o2[:K + 1].copy(), threads=os.cpu_count() or 8)
    t_cpu = time.time() - t0
    bad = {}
    def chk(name, x, y):
        n = int((np.asarray(x) != np.asarray(y)).sum())
        if n: bad[name] = n
    chk("paired", ores["paired"], out["paired"][:K])
    nc_ = kw["v"] + 1
    chk("n_pairs", ores["n_pairs"][:, :2 * nc_ - 1], npairs[:K, :2 * nc_ - 1])
    up = (ores["tmp"] == 1) | (ores["paired"] == 0)
    chk("unpaired_out", up, out["unpaired_out"][:K] != 0)
    pr = ~up
    for fld in ("chain", "na", "nb", "insert", "a_chr", "a_loc", "b_chr", "b_loc"):
        chk("pick." + fld, ores["pick"][fld][pr], out[fld][:K][pr])
    for m_, cnts in (("a", ca), ("b", cb)):
        ok = ores[m_]["filtered"] == 0
        chk(m_ + ".n_hit", ores[m_]["n_hit"][ok][:, :nc_], cnts["n_hit"][:K][ok][:, :nc_])
        chk(m_ + ".n_chit", ores[m_]["n_chit"][ok][:, :nc_], cnts["n_chit"][:K][ok][:, :nc_])
        sel = up & ok & (ores[m_]["n_best"] > 0)
        for fld in ("chr", "loc", "best_class"):
            chk(f"{m_}.{fld}", ores[m_][fld][sel], out[m_][fld][:K][sel])
    load = ca["n_hit"][:K].sum(1).astype(np.int64) + cb["n_chit"][:K].sum(1)
    print(json.dumps({"pairs_on_gpu": a.pairs, "pairs_checked": K, "paired": int(pr.sum()), "heavy_units_in_batch": int(pa.heavy_units()),
                      "max_hits_in_checked_unit": int(load.max()), "oracle_s": round(t_cpu, 1), "options": kw,
                      "genome_seed": a.genome_seed, "read_seed": a.read_seed, "mismatching_fields": bad}))
    pa.close(); ref.close()
    sys.exit(1 if bad else 0)


def validate_se(a, kw):
    ref = B.RefSeq(B.make_params(**kw)).synthetic(HG38, seed=a.genome_seed).CreateIndex()
    sa = B.SingleAlign(ref, a.pairs)
    sa.synth_reads(a.pairs, 100, seed=a.read_seed)
    sa.Do_Batch()
    hits, cc = sa.results()
    K = min(a.check, a.pairs)
    f, c = ref.words(); an, sz, rc = ref.info(); off, nf, ent = ref.index()
    oref = O.OracleRef.wrap(O.make_params(**kw), f, c, an, sz, rc, off, nf, ent)
    b1, o1 = sa.download_reads(0)
    t0 = time.time()
    ores, ocnt = O.se_batch(oref, b1[:int(o1[K])], o1[:K + 1].copy(), threads=os.cpu_count() or 8)
    t_cpu = time.time() - t0
    nc_ = kw["v"] + 1
    bad = {}
    for name, x, y in (("n_hit", ores["n_hit"][:, :nc_], cc["n_hit"][:K, :nc_]), ("n_chit", ores["n_chit"][:, :nc_], cc["n_chit"][:K, :nc_]),
                       ("n_best", np.maximum(ores["n_best"], 0), hits["n_best"][:K])):
        n = int((np.asarray(x) != np.asarray(y)).sum())
        if n: bad[name] = n
    has = ores["n_best"] > 0
    for fld in ("chr", "loc", "best_class"):
        n = int((ores[fld][has] != hits[fld][:K][has]).sum())
        if n: bad[fld] = n
    print(json.dumps({"reads_on_gpu": a.pairs, "reads_checked": K, "placed": int(has.sum()), "heavy_units_in_batch": int(sa.heavy_units()), "oracle_s": round(t_cpu, 1),
                      "options": kw, "genome_seed": a.genome_seed, "read_seed": a.read_seed, "mismatching_fields": bad}))
    sa.close(); ref.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
