#!/usr/bin/env python3
"""Golden SAM of the REAL `bsmap -p 1` (oracle/_ref/bsmap) for inputs whose planner state travels far: one read that sets the
start offset, then runs of hundreds of consecutive reads with (len - I + 1) % S == 0 (they never set it, align.cpp:458-468) whose
tail entries of seed_array were written by a long read far behind them.  The command-line test cuts these inputs into small
batches, so the state has to cross several batch boundaries.  Run in the build container only; stores reads, options and the
output text."""
import gzip
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bsx_testdata as td  # noqa: E402
from oracle import ref_ffi as R  # noqa: E402

FASTA = os.path.join(HERE, "genome_wgbs.fa")


def genome():
    out, name, seq = [], None, []
    for ln in open(FASTA):
        if ln.startswith(">"):
            if name:
                out.append((name, "".join(seq)))
            name, seq = ln[1:].split()[0], []
        else:
            seq.append(ln.strip())
    out.append((name, "".join(seq)))
    return out


def lengths(rng, n):
    """a long non-leaky read first, then runs of leaky lengths (99, 83, 67, 51 with -s 16 -I 4) broken by a few other reads"""
    ln = [100, 97]
    while len(ln) < n:
        ln += [int(rng.choice([99, 83, 67, 51]))] * int(rng.integers(90, 260))
        ln += [int(x) for x in rng.choice([100, 96, 90, 77, 60, 99], int(rng.integers(1, 4)))]
    return ln[:n]


def main():
    assert R.build()
    g = genome()
    tmp = tempfile.mkdtemp()
    rng = np.random.default_rng(77)
    out = {}
    # single-end, short reads with a few mismatches (C1-like: two seeds, -v 2): which seeds the planner picks decides whether a hit is found,
    # so the state a leaky read inherits shows in the output
    from oracle import oracle_ffi as O
    n = 900
    reads = td.make_se_reads(g, n, 39, seed=31, junk_frac=0.0, sub_rate=0.045)
    ln = [39 - 1, 36]   # (len - 4 + 1) % 12 != 0: these set the offset
    while len(ln) < n:
        ln += [int(rng.choice([39, 27]))] * int(rng.integers(90, 260))   # never set it
        ln += [int(x) for x in rng.choice([38, 36, 33, 30], int(rng.integers(1, 3)))]
    for r, L in zip(reads, ln[:n]):
        r["seq"], r["qual"] = r["seq"][:L], r["qual"][:L]
    f1 = os.path.join(tmp, "se.fq")
    with open(f1, "w") as a:
        for r in reads:
            a.write(f"@{r['name']}\n{r['seq']}\n+\n{r['qual']}\n")
    kw = dict(s=12, v=2, I=4, S=1, r=1, n=1)
    opts = ["-s", "12", "-v", "2", "-I", "4", "-S", "1", "-r", "1", "-n", "1", "-R", "-u"]
    o = os.path.join(tmp, "se.sam")
    R.run_bsmap(["-a", f1, "-d", FASTA, "-o", o, "-p", "1"] + opts)
    # the input must tell the modes apart: the oracle with and without the running state
    oref = O.OracleRef(O.make_params(**kw), fasta_path=FASTA)
    buf, off = O.pack_reads([r["seq"] for r in reads])
    e1, _ = O.se_batch(oref, buf, off, threads=1, leak_mode=1)
    e0, _ = O.se_batch(oref, buf, off, threads=1, leak_mode=0)
    differ = int(((e1["n_best"] != e0["n_best"]) | (e1["loc"] != e0["loc"]) | (e1["chr"] != e0["chr"])).sum())
    print("se: reads whose reported hit depends on the running state:", differ)
    assert differ >= 10
    out["se_n1"] = dict(kind="se", options=opts, reads=[dict(name=r["name"], seq=r["seq"], qual=r["qual"]) for r in reads], out=open(o).read(), differ=differ)
    # paired-end: each mate stream carries its own state
    n = 500
    pairs = td.make_pe_reads(g, n, 144, seed=32)
    la = [int(x) for x in np.where(rng.random(n) < 0.9, rng.choice([131, 115, 99], n), rng.choice([144, 140, 120], n))]
    lb = [int(x) for x in np.where(rng.random(n) < 0.9, rng.choice([131, 115], n), rng.choice([144, 101], n))]
    la[0], lb[0] = 144, 144
    for p, x, y in zip(pairs, la, lb):
        p["seq1"], p["qual1"], p["seq2"], p["qual2"] = p["seq1"][:x], p["qual1"][:x], p["seq2"][:y], p["qual2"][:y]
    f1, f2 = os.path.join(tmp, "pe_1.fq"), os.path.join(tmp, "pe_2.fq")
    with open(f1, "w") as a, open(f2, "w") as b:
        for p in pairs:
            a.write(f"@{p['name']}/1\n{p['seq1']}\n+\n{p['qual1']}\n")
            b.write(f"@{p['name']}/2\n{p['seq2']}\n+\n{p['qual2']}\n")
    opts = ["-s", "16", "-v", "6", "-I", "4", "-S", "1", "-r", "1", "-m", "28", "-x", "500", "-R", "-u"]
    o = os.path.join(tmp, "pe.sam")
    R.run_bsmap(["-a", f1, "-b", f2, "-d", FASTA, "-o", o, "-p", "1"] + opts)
    out["pe"] = dict(kind="pe", options=opts, reads=[dict(name=p["name"], seq1=p["seq1"], qual1=p["qual1"], seq2=p["seq2"], qual2=p["qual2"]) for p in pairs],
                     out=open(o).read())
    json.dump(out, gzip.open(os.path.join(HERE, "cli_leakrun.json.gz"), "wt"))
    print({k: v["out"].count("\n") for k, v in out.items()})


if __name__ == "__main__":
    main()
