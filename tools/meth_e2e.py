#!/usr/bin/env python3
"""Wall time of `bsmap_amd.methratio` on a BSP file (parse in Python, pile-up and row selection on the GPU, table written
from Python) — the same kind of input the reference script was timed on in the build container:
4 Mb genome, 200 000 × 100 nt reads mapped by bsmap_amd/bsmap.  Prints one JSON line."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bsx_testdata as td
from bsmap_amd import methratio


def main():
    tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    g = td.make_genome(seed=3, chr_lens=(3_000_000, 1_000_000), gc=0.45, repeats=50, microsats=20)
    fa = os.path.join(tmp, "g.fa"); td.write_fasta(fa, g)
    reads = td.make_se_reads(g, 200000, 100, seed=5)
    fq = os.path.join(tmp, "r.fq"); td.write_fastq(fq, reads)
    bsp = os.path.join(tmp, "o.bsp")
    subprocess.run([os.path.join(ROOT, "bsmap_amd", "bsmap"), "-a", fq, "-d", fa, "-o", bsp, "-s", "16", "-v", "4", "-S", "1"], check=True, capture_output=True)
    n = sum(1 for _ in open(bsp))
    methratio.run(fa, [bsp], os.path.join(tmp, "warm.txt"))
    t0 = time.time()
    s = methratio.run(fa, [bsp], os.path.join(tmp, "m.txt"))
    dt = time.time() - t0
    rows = sum(1 for _ in open(os.path.join(tmp, "m.txt"))) - 1
    print(json.dumps({"alignment_lines": n, "wall_s": round(dt, 2), "lines_per_s": round(n / dt), "table_rows": rows, "summary": s.strip()}))


if __name__ == "__main__":
    main()
