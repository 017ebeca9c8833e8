"""RRBS at the bench scale: the duplicate-suppression set of single-end RRBS units must take the reads that match thousands of
places (a poly-T read of the hg38-sized genome used to write past its slab)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bsmap_amd as B, bench
kw = dict(D="C-CGG", S=1, r=1)
ref = B.RefSeq(B.make_params(**kw)).synthetic(bench.HG38, seed=38).CreateIndex()
n = 1 << 21
sa = B.SingleAlign(ref, n)
sa.synth_reads(n, 75, seed=3, kind=2)
sa.run_range(0, n, sync=True)
h, cc = sa.results()
tot = cc["n_hit"].sum(1).astype(np.int64) + cc["n_chit"].sum(1)
print(json.dumps({"n": n, "placed": float((h["n_best"] > 0).mean()), "flag_limit": int(((h["flags"] & 4) != 0).sum()), "max_hits_per_read": int(tot.max()),
                  "reads_with_1000_hits": int((tot >= 1000).sum()), "kernel_ms": sa.kernel_ms()}))
