#!/usr/bin/env python3
"""A whole bench-sized batch against the oracle (GPU box, outside the suite: the oracle needs a minute or two for 2^20 units): the
hg38-sized synthetic genome, one Do_Batch over --units units of a BASELINE config, EVERY unit re-aligned by the oracle's batch driver
on the host cores against the same reference + index copied back from HBM — every record field and the four work counters
(tests/wholebatch.py, the comparison the -m gpu suite runs on 65 536-131 072 units).  Writes profiles-style JSON to stdout.
usage: validate_fullsize.py [--mode pe|se|trim] [--units 1048576] [--exact]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench as BN
import bsmap_amd as B
import wholebatch as W
from oracle import oracle_ffi as O


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="pe", choices=["pe", "se", "trim"])
    ap.add_argument("--units", type=int, default=1 << 20)
    ap.add_argument("--read-seed", type=int, default=3)
    ap.add_argument("--exact", action="store_true")
    a = ap.parse_args()
    M = BN.MODES[a.mode]
    kw = M["kw"]
    ref = B.RefSeq(B.make_params(**kw)).synthetic(BN.HG38, seed=38).CreateIndex()
    al = (B.PairAlign if M["pe"] else B.SingleAlign)(ref, a.units)
    if a.exact:
        al.set_leak_exact()
    al.synth_reads(a.units, M["L"], seed=a.read_seed, kind=M["kind"])
    al.Do_Batch()
    res = al.results()
    cnt = [int(x) for x in al.counters()[:4]]
    f, c = ref.words(); an, sz, rc = ref.info(); off, nf, ent = ref.index()
    oref = O.OracleRef.wrap(O.make_params(**kw), f, c, an, sz, rc, off, nf, ent)
    ores, ocnt, t_cpu = W.run_oracle(O, oref, al, M["pe"], M["kind"] == 1, a.units, leak_mode=1 if a.exact else 0)
    nclass = kw.get("v", 2) + 1
    bad, info = (W.compare_pe(ores, res[0], res[1], res[2], res[3], nclass) if M["pe"] else W.compare_se(ores, res[0], res[1], nclass))
    out = dict(info, config=M["tag"], units=a.units, exact_mode=bool(a.exact), oracle_s=round(t_cpu, 1), oracle_threads=W.usable_cpus(), counters_gpu=cnt, counters_oracle=ocnt,
               mismatching_fields=bad, heavy_units=int(al.heavy_units()), lib_sha16=BN.lib_sha16(), read_seed=a.read_seed,
               options={k: v for k, v in kw.items()})
    print(json.dumps(out))
    al.close(); ref.close()
    sys.exit(1 if (bad or cnt != ocnt) else 0)


if __name__ == "__main__":
    main()
