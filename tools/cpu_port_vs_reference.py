#!/usr/bin/env python3
"""Build container only: is the oracle (the `cpu_baseline.kind = "port"` of bench.py) a fair stand-in for the real
reference on the CPU?  Same genome, same reads, same thread count: the real `bsmap -p N` (oracle/_ref/bsmap, mapping
phase = total minus the time to the "Create seed table" line, 1-second resolution) against the oracle's pthread batch
driver.  Prints one JSON object.  usage: cpu_port_vs_reference.py [--mb 30] [--reads 400000] [--threads 8]"""
import argparse, json, os, re, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bsx_testdata as td
from oracle import oracle_ffi as O, ref_ffi as R


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=float, default=30)
    ap.add_argument("--reads", type=int, default=400000)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--density", type=float, default=0.1, help="--pe: microsatellite / repeat density relative to the parity suite's heavy genome (a third microsatellite)")
    ap.add_argument("--pe", action="store_true", help="the regime of the bench's baseline instead (VERDICT r4 #5): 2x144 nt pairs, -v 6 -m 28 -x 500, on a genome a third of which is "
                    "microsatellite (thousands of candidates per read: the reference's std::set inserts and sorts of 1000-hit lists weigh in)")
    a = ap.parse_args()
    if a.pe:
        return main_pe(a)
    assert R.build()
    tmp = tempfile.mkdtemp()
    n = int(a.mb * 1e6)
    g = td.make_genome(seed=9, chr_lens=(n // 2, n - n // 2), gc=0.41, repeats=int(40 * a.mb), microsats=int(30 * a.mb), n_runs=8)
    fa = os.path.join(tmp, "g.fa"); td.write_fasta(fa, g)
    reads = td.make_se_reads(g, a.reads, 100, seed=4)
    fq = os.path.join(tmp, "r.fq")
    with open(fq, "w") as f:
        for r in reads:
            f.write(f"@{r['name']}\n{r['seq']}\n+\n{r['qual']}\n")
    kw = dict(s=16, v=4, I=4, S=1, r=1)
    t0 = time.time()
    out = R.run_bsmap(["-a", fq, "-d", fa, "-o", os.path.join(tmp, "o.sam"), "-s", 16, "-v", 4, "-I", 4, "-S", 1, "-p", a.threads])
    t_ref_wall = time.time() - t0
    seed_s = int(re.search(r"Create seed table\. (\d+) secs passed", out).group(1))
    total_s = int(re.search(r"Total time consumed:\s+(\d+) secs", out).group(1))
    oref = O.OracleRef(O.make_params(**kw), fasta_path=fa)
    sb, so = O.pack_reads([r["seq"] for r in reads])
    t0 = time.time()
    res, cnt = O.se_batch(oref, sb, so, threads=a.threads)
    t_port = time.time() - t0
    print(json.dumps({"genome_mb": a.mb, "reads": a.reads, "threads": a.threads,
                      "reference": {"mapping_s": total_s - seed_s, "index_s": seed_s, "wall_s": round(t_ref_wall, 1),
                                    "reads_per_s": round(a.reads / max(1, total_s - seed_s))},
                      "port": {"mapping_s": round(t_port, 2), "reads_per_s": round(a.reads / t_port)},
                      "aligned_port": int((res["n_best"] > 0).sum())}))


def main_pe(a):
    assert R.build()
    tmp = tempfile.mkdtemp()
    n = int(a.mb * 1e6)
    # (the density of tests/test_gpu_parity.py's heavy genome: 3000 microsatellites and 200 repeat copies per 2.5 Mb)
    g = td.make_genome(seed=5, chr_lens=(n * 4 // 5, n - n * 4 // 5), gc=0.45, microsats=int(1200 * a.mb * a.density), repeats=int(80 * a.mb * a.density), n_runs=4)
    fa = os.path.join(tmp, "g.fa"); td.write_fasta(fa, g)
    pairs = td.make_pe_reads(g, a.reads, 144, seed=8, sub_rate=0.01)
    fq1, fq2 = os.path.join(tmp, "r1.fq"), os.path.join(tmp, "r2.fq")
    with open(fq1, "w") as f1, open(fq2, "w") as f2:
        for p_ in pairs:
            f1.write(f"@{p_['name']}/1\n{p_['seq1']}\n+\n{'I' * len(p_['seq1'])}\n")
            f2.write(f"@{p_['name']}/2\n{p_['seq2']}\n+\n{'I' * len(p_['seq2'])}\n")
    kw = dict(s=16, v=6, I=4, m=28, x=500, S=1, r=1, pairend=1)
    oref = O.OracleRef(O.make_params(**kw), fasta_path=fa)
    s1, o1 = O.pack_reads([p_["seq1"] for p_ in pairs])
    s2, o2 = O.pack_reads([p_["seq2"] for p_ in pairs])
    t0 = time.time()
    res, cnt = O.pe_batch(oref, s1, o1, s2, o2, threads=a.threads)
    t_port = time.time() - t0
    print("port: %.1f s, %.0f candidates per read" % (t_port, cnt[1] / (2.0 * a.reads)), file=sys.stderr, flush=True)
    t0 = time.time()
    out = R.run_bsmap(["-a", fq1, "-b", fq2, "-d", fa, "-o", os.path.join(tmp, "o.sam"), "-s", 16, "-v", 6, "-I", 4, "-m", 28, "-x", 500, "-S", 1, "-p", a.threads])
    t_ref_wall = time.time() - t0
    seed_s = int(re.search(r"Create seed table\. (\d+) secs passed", out).group(1))
    total_s = int(re.search(r"Total time consumed:\s+(\d+) secs", out).group(1))
    print(json.dumps({"regime": "PE 2x144, -v 6 -m 28 -x 500, repeat-rich genome", "genome_mb": a.mb, "pairs": a.reads, "threads": a.threads,
                      "candidates_per_read": round(cnt[1] / (2.0 * a.reads), 1), "lookups_per_read": round(cnt[0] / (2.0 * a.reads), 1),
                      "reference": {"mapping_s": total_s - seed_s, "index_s": seed_s, "wall_s": round(t_ref_wall, 1), "reads_per_s": round(2 * a.reads / max(1, total_s - seed_s)),
                                    "note": "the real bsmap's own clock has 1-second resolution"},
                      "port": {"mapping_s": round(t_port, 2), "reads_per_s": round(2 * a.reads / t_port)},
                      "port_over_reference": round((total_s - seed_s) / t_port, 2), "paired_port": int((res["paired"] > 0).sum())}))


if __name__ == "__main__":
    main()
