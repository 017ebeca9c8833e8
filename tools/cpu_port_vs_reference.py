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
    a = ap.parse_args()
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


if __name__ == "__main__":
    main()
