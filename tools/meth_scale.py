#!/usr/bin/env python3
"""bsmap_amd.methratio at the bench scale: the SAM file of tools/e2e_bench.py (N pairs against the hg38-sized synthetic
genome) -> methylation table.  Prints one JSON line with the phase times.  usage: meth_scale.py [--pairs 4194304]"""
import argparse, json, os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import e2e_bench
from bsmap_amd import methratio


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=4 << 20)
    a = ap.parse_args()
    d = "/dev/shm/bsx_meth_scale"
    r = e2e_bench._run(types.SimpleNamespace(pairs=a.pairs, genome=1.0, dir=d, threads=0, keep=True))
    try:
        t0 = time.time()
        s = methratio.run(os.path.join(d, "genome.fa"), [os.path.join(d, "out.sam")], os.path.join(d, "meth.txt"), quiet=False)
        dt = time.time() - t0
        rows = sum(1 for _ in open(os.path.join(d, "meth.txt"))) - 1
        print(json.dumps({"pairs": a.pairs, "sam_bytes": r["sam_bytes"], "methratio_wall_s": round(dt, 2), "alignments_per_s": round(2 * a.pairs / dt), "table_rows": rows,
                          "table_bytes": os.path.getsize(os.path.join(d, "meth.txt")), "summary": s.strip()}))
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
