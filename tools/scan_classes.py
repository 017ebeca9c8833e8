#!/usr/bin/env python3
"""Share of the scan kernel's candidates by the reference's early-out class (1, 2 or 5 words): decides what the tail of
k_hscan is worth.  GPU box only."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import numpy as np
import bsmap_amd as B
import bench
kw = dict(s=16, v=6, I=4, m=28, x=500, S=1, r=1, pairend=1)
ref = B.RefSeq(B.make_params(**kw)).synthetic(bench.HG38, seed=38)
ref.CreateIndex()
n = 1 << 20
pa = B.PairAlign(ref, n)
pa.synth_reads(n, 144, seed=3)
pa.run_range(0, n, sync=True)
c = pa.counters().astype(np.float64)
cand, words, n1, n5 = c[7], c[8], c[9], c[10]
n2 = cand - n1 - n5
print(json.dumps({"scan_candidates": cand, "words_per_candidate": words / cand, "class_one": n1 / cand, "class_two": n2 / cand, "class_five": n5 / cand,
                  "all_counters": [float(x) for x in c]}))
