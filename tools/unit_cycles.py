import sys, time, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import os
import bsmap_amd as B
if os.environ.get("BSXLIB"): B.LIB_PATH=os.environ["BSXLIB"]
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
HG38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]
lens = [max(200000, int(x * frac)) for x in HG38]
kw = dict(s=16, v=6, I=4, m=28, x=500, S=1, r=1, pairend=1)
TRIM = os.environ.get("MODE") == "trim"  # C5: -q 20 -A, reads with low-quality tails and adapter read-through
if TRIM: kw.update(q=20, A=["AGATCGGAAGAGC"])
ref = B.RefSeq(B.make_params(**kw)).synthetic(lens, seed=38).CreateIndex()
pa = B.PairAlign(ref, n); pa.synth_reads(n, 144, seed=3, kind=1 if TRIM else 0)
pa.Do_Batch(); pa.Do_Batch(); print("kernel ms", pa.kernel_ms())
pa.set_debug(2); pa.Do_Batch(); print("kernel ms (cycles on)", pa.kernel_ms())
c = pa.unit_cycles().astype(np.float64)
out, ca, cb, npairs = pa.results()
print("cycles: mean %.0f median %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f  sum/1e9 %.2f" % (c.mean(), np.median(c), *np.percentile(c, [90, 99, 99.9]), c.max(), c.sum() / 1e9))
heavy = np.argsort(c)[-5:]
for i in heavy: print(" unit", i, "cycles", c[i], "paired", out[i]["paired"], "n_pairs", out[i]["n_pairs"], "a", ca[i]["n_hit"][:7], ca[i]["n_chit"][:7], "b", cb[i]["n_hit"][:7], cb[i]["n_chit"][:7])
tot = c.sum(); srt = np.sort(c)[::-1]; print("share of cycles in top 1%% units: %.2f, top 5%%: %.2f" % (srt[:n // 100].sum() / tot, srt[:n // 20].sum() / tot))
print("counters", pa.counters())
print("heavy units", pa.heavy_units())

cc = pa.ctrl_clocks().astype(float); print("ctrl clocks (Mcycles): prepare/restore %.0f inline-scan %.0f replay %.0f sort+pairs %.0f save/finish %.0f recount %.0f advance-total %.0f" % tuple(cc[[0,1,2,3,4,5,6]]/1e6))
print("longest single span (kcycles): prepare/restore %.0f inline-scan %.0f replay %.0f sort+pairs %.0f save/finish %.0f recount %.0f advance %.0f" % tuple(cc[[8,9,10,11,12,13,14]]/1e3))
print("longest visit break-down (kcycles x spans): " + " ".join("%s %.0f x%d" % (n, (int(v) >> 16) / 1e3, int(v) & 0xffff) for n, v in zip(["restore", "inline-scan", "replay", "sort+pairs", "save", "recount"], cc[16:22])))
