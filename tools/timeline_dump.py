#!/usr/bin/env python3
"""first N kernel launches of the last k_align-started step of a rocprofv3 --kernel-trace csv: start offset (us), duration (us), name"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(f))))
starts = [i for i, r in enumerate(rows) if "k_align" in r[2]]
a = starts[-2] if len(starts) > 1 else starts[-1]
t0 = rows[a][0]
prev_end = t0
for s, e, k, q in rows[a + skip:a + skip + n]:
    short = k.split("(")[0].split("::")[-1][:28]
    print("%10.1f  +%8.1f  gap %8.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, q, short))
    prev_end = max(prev_end, e)

big = sorted(((e - s, k) for s, e, k, q in rows if not any(x in k for x in ("k_align", "k_hctrl", "k_hscan"))), reverse=True)[:8]
for d, k in big: print("other %8.1f us  %s" % (d / 1e3, k[:120]))
