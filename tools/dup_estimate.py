#!/usr/bin/env python3
"""Planning diagnostic (GPU box): how much of the scan work of the heaviest reads is the SAME bucket visited again at
another seed level with a congruent offset (DESIGN.md §7 "cross-level reuse")?  For mate 1 of every pair of a batch the
bucket of every (level, phase) lookup is recomputed from the planner state (bsx_batch_debug_plan) and the lookups are
grouped by (bucket, read offset mod I).  Prints one JSON line."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bsmap_amd as B

HG38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
        135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
        46709983, 50818468, 156040895, 57227415]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    kw = dict(s=16, v=6, I=4, m=28, x=500, S=1, r=1, pairend=1)
    p = B.make_params(**kw)
    ref = B.RefSeq(p).synthetic(HG38, seed=38).CreateIndex()
    pa = B.PairAlign(ref, n, debug=True)
    pa.synth_reads(n, 144, seed=77)
    pa.Do_Batch()
    out, ca, cb, npairs = pa.results()
    off = ref.index()[0].astype(np.int64)
    b1, o1 = pa.download_reads(0)
    code = np.zeros(256, np.int64)
    for ch, k in zip("ACGT", range(4)):
        code[ord(ch)] = p.bit_nt[k]; code[ord(ch.lower())] = p.bit_nt[k]
    S, I = p.seed_size, p.index_interval
    pw = 3 ** np.arange(S - 1, -1, -1)
    tot_all = dup_all = 0
    tot_heavy = dup_heavy = n_heavy = 0
    for u in range(n):
        seq = code[b1[int(o1[u]):int(o1[u + 1])]]
        t3 = np.where(seq == 3, 1, seq)  # the read's T reads as C in the 3-letter hash
        starts, order = pa.debug_plan(u, 0)
        levels = int(out["paired"][u]) if out["paired"][u] else 7
        groups = {}
        tot = 0
        for L in range(min(levels, 7)):
            seg = int(order[0][L])
            for ph in range(I):
                o = int(p.profile_a[seg][ph]) + int(starts[0][seg]) - ph
                if o < 0 or o + S > len(t3):
                    continue
                key = int((t3[o:o + S] * pw).sum())
                size = int(off[key + 1] - off[key])
                tot += size
                groups.setdefault((key, o % I), []).append(size)
        dup = sum((len(v) - 1) * v[0] for v in groups.values())
        tot_all += tot; dup_all += dup
        if tot >= 32768:
            n_heavy += 1; tot_heavy += tot; dup_heavy += dup
    print(json.dumps({"units": n, "candidates_mate1": tot_all, "repeat_visits": dup_all, "fraction": round(dup_all / max(1, tot_all), 3),
                      "heavy_units": n_heavy, "heavy_candidates": tot_heavy, "heavy_repeat_visits": dup_heavy,
                      "heavy_fraction": round(dup_heavy / max(1, tot_heavy), 3), "heavy_share_of_all": round(tot_heavy / max(1, tot_all), 3)}))
    pa.close(); ref.close()


if __name__ == "__main__":
    main()
