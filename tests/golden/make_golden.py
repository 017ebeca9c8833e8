#!/usr/bin/env python3
"""Generates the golden vectors in this directory from the REAL reference (BSMAP v2.6).

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py
It compiles the reference where it lies (make -C oracle ref -> oracle/_ref/) and records, for small
seeded inputs shaped like BASELINE.json's five configs:
  * the inputs (genome FASTA text, reads),
  * white-box internals of the reference objects (packed reference words, anchors, blocks, the
    non-empty seed-index buckets, per-read planner state, every hit list, every pair list),
  * the text the reference's own formatter produced per read, and the output file of the real
    `bsmap` binary run with -p 1 on the same files.
Only data is stored here; no reference source text.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bsx_testdata as td  # noqa: E402
from oracle import ref_ffi as R  # noqa: E402

ADAPTER = "AGATCGGAAGAGC"
CONFIGS = {
    # C1: 1x36 bp, -s 12 -v 2 (plumbing config)
    "c1_se36": dict(kind="se", kw=dict(s=12, v=2, I=4, S=1, r=1), n=400, length=36, sub=0.02, strands=("++", "-+")),
    # C2: 1x100 bp WGBS -s 16 -v 4 -I 4; also all four strands (-n 1) and unique-only (-r 0)
    "c2_se100": dict(kind="se", kw=dict(s=16, v=4, I=4, S=1, r=1), n=400, length=100, sub=0.01, strands=("++", "-+")),
    "c2_se100_n1": dict(kind="se", kw=dict(s=16, v=4, I=4, S=2, r=1, n=1), n=400, length=100, sub=0.01, strands=("++", "-+", "+-", "--")),
    "c2_se100_r0_w3": dict(kind="se", kw=dict(s=16, v=4, I=4, S=2, r=0, n=1, w=3), n=400, length=100, sub=0.01, strands=("++", "-+", "+-", "--")),
    # C3: 2x150 (->144) PE -s 16 -v 6 -m 28 -x 500
    "c3_pe150": dict(kind="pe", kw=dict(s=16, v=6, I=4, S=1, r=1, m=28, x=500), n=300, length=150, sub=0.01),
    "c3_pe150_r0": dict(kind="pe", kw=dict(s=16, v=6, I=4, S=1, r=0, m=28, x=500), n=300, length=150, sub=0.01),
    # C4: RRBS -D C-CGG 1x75
    "c4_rrbs75": dict(kind="se", kw=dict(D="C-CGG", v=2, S=1, r=1), n=400, length=75, sub=0.005, rrbs=True),
    # C5: adapter + quality trimming PE
    "c5_trim_pe150": dict(kind="pe", kw=dict(s=16, v=6, I=4, S=1, r=1, m=28, x=500, q=20, A=[ADAPTER]), n=300, length=150, sub=0.01, trim=True),
}


def state_dict(st):
    d = {}
    for f, _ in st._fields_:
        v = getattr(st, f)
        d[f] = list(v) if hasattr(v, "__len__") else v
    return d


def main():
    assert R.build(), "needs /root/reference"
    g_w = td.make_genome(seed=11, chr_lens=(40_000, 25_000, 9_013), gc=0.5, n_runs=5, repeats=12, microsats=8, lower=3, iupac=4)
    g_r = td.make_genome(seed=12, chr_lens=(60_000, 20_000), gc=0.55, n_runs=3, repeats=5, microsats=3, cpg_sites=300)
    open(os.path.join(HERE, "genome_wgbs.fa"), "w").write(td.fasta_text(g_w))
    open(os.path.join(HERE, "genome_rrbs.fa"), "w").write(td.fasta_text(g_r))
    tmp = tempfile.mkdtemp()
    for name, cfg in CONFIGS.items():
        kw = dict(cfg["kw"], out_sam=1)
        rrbs = cfg.get("rrbs", False)
        genome = g_r if rrbs else g_w
        fa = os.path.join(HERE, "genome_rrbs.fa" if rrbs else "genome_wgbs.fa")
        if cfg["kind"] == "pe":
            kw["pairend"] = 1
        ref = R.Reference(fa, **kw)
        arrays = dict(refcat=ref.refcat()[400:-400].copy(), crefcat=ref.crefcat()[400:-400].copy(), anchor=ref.anchor(),
                      chr_size=ref.chr_size(), rc_offset=ref.rc_offset(), blocks=ref.blocks())
        if rrbs:
            off, ent = ref.rrbs_csr()
            cnt = np.diff(off.astype(np.int64))
            keys = np.nonzero(cnt)[0].astype(np.uint32)
            arrays.update(idx_keys=keys, idx_n=cnt[keys].astype(np.uint32), idx_entries=ent,
                          sites0=ref.sites(0), sites1=ref.sites(1))
        else:
            off, nfw, ent = ref.csr()
            cnt = np.diff(off.astype(np.int64))
            keys = np.nonzero(cnt)[0].astype(np.uint32)
            arrays.update(idx_keys=keys, idx_n=cnt[keys].astype(np.uint32), idx_nfwd=nfw[keys], idx_entries=ent)
        nclass = kw["v"] + 1
        recs = []
        if cfg["kind"] == "se":
            if rrbs:
                reads = td.make_rrbs_reads(genome, cfg["n"] - 60, cfg["length"], seed=21) + td.make_se_reads(genome, 60, cfg["length"], seed=22, var_len=True)
            else:
                reads = td.make_se_reads(genome, cfg["n"], cfg["length"], seed=21, sub_rate=cfg["sub"], strands=cfg["strands"])
                reads += td.make_se_reads(genome, 40, cfg["length"], seed=23, sub_rate=0.03, strands=cfg["strands"], var_len=True)
            for i, r in enumerate(reads):
                st, line = ref.se(i, r["name"], r["seq"], r["qual"])
                d = state_dict(st)
                d["line"] = line
                d["hits"] = [[ref.se_hits(o, w, (st.n_chit if o else st.n_hit)[w]) for o in (0, 1)] for w in range(nclass)] if not st.filtered else []
                recs.append(d)
            fq = os.path.join(tmp, name + ".fq")
            td.write_fastq(fq, reads)
            out = os.path.join(tmp, name + ".sam")
            args = ["-a", fq, "-d", fa, "-o", out, "-p", 1]
        else:
            trim = cfg.get("trim", False)
            pairs = td.make_pe_reads(genome, cfg["n"], cfg["length"], seed=31, sub_rate=cfg["sub"], qual_tail=trim,
                                     adapter=ADAPTER if trim else None, ins_min=30 if trim else 50, ins_mean=250 if trim else 300,
                                     ins_sd=90 if trim else 50)
            reads = pairs
            for i, r in enumerate(pairs):
                st, l1, l2 = ref.pe(i, r["name"] + "/1", r["seq1"], r["qual1"], r["name"] + "/2", r["seq2"], r["qual2"])
                d = dict(paired=st.paired, tmp=st.tmp, n_pairs=list(st.n_pairs), a=state_dict(st.a), b=state_dict(st.b), line=l1)
                for mate, x in enumerate((st.a, st.b)):
                    d["ab"[mate]]["hits"] = [[ref.pe_hits(mate, o, w, (x.n_chit if o else x.n_hit)[w]) for o in (0, 1)] for w in range(nclass)] if not x.filtered else []
                d["pairs"] = [ref.pe_pairs(w, st.n_pairs[w]) for w in range(2 * nclass - 1)] if (not st.a.filtered and not st.b.filtered) else []
                recs.append(d)
            fq1, fq2 = os.path.join(tmp, name + "_1.fq"), os.path.join(tmp, name + "_2.fq")
            for r in pairs:
                r["n1"], r["n2"] = r["name"] + "/1", r["name"] + "/2"
            with open(fq1, "w") as f:
                for r in pairs:
                    f.write(f"@{r['n1']}\n{r['seq1']}\n+\n{r['qual1']}\n")
            with open(fq2, "w") as f:
                for r in pairs:
                    f.write(f"@{r['n2']}\n{r['seq2']}\n+\n{r['qual2']}\n")
            out = os.path.join(tmp, name + ".sam")
            args = ["-a", fq1, "-b", fq2, "-d", fa, "-o", out, "-p", 1]
        for k, v in cfg["kw"].items():
            if k == "A":
                for a in v:
                    args += ["-A", a]
            elif k == "D":
                args = ["-D", v] + args  # -D must precede -s/-I (main.cpp:247,257)
            else:
                args += ["-" + k, v]
        R.run_bsmap(args)
        sam = open(out).read()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
        import gzip
        json.dump(dict(config=name, kind=cfg["kind"], kw=kw, reads=reads, expected=recs, bsmap_args=[str(a) for a in args[6 if cfg['kind']=='se' else 8:]],
                       bsmap_sam=sam), gzip.open(os.path.join(HERE, name + ".json.gz"), "wt"))
        print(name, "reads", len(reads), "sam lines", sam.count("\n"))


if __name__ == "__main__":
    main()
