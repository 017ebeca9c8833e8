#!/usr/bin/env python3
"""print per-kernel totals per Do_Batch from a rocprofv3 --kernel-trace csv directory"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = {}; n_main = 0
for r in rows:
    k = r["Kernel_Name"]
    short = "k_align" if "k_align" in k else "k_hctrl" if "k_hctrl" in k else "k_hscan" if "k_hscan" in k else None
    if not short: continue
    if short == "k_align": n_main += 1
    d = tot.setdefault(short, [0, 0.0]); d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
ts = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if any(x in r["Kernel_Name"] for x in ("k_align", "k_hctrl", "k_hscan"))]
print("steps", n_main, {k: (v[0] / n_main, round(v[1] / n_main, 2)) for k, v in tot.items()}, "sum ms/step", round(sum(v[1] for v in tot.values()) / n_main, 2))
