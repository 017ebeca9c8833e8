#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per kernel: usage pmc_table.py <dir> [<dir> ...]"""
import csv, glob, os, sys, collections
for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, c in acc.items():
        if not any(t in k for t in ("k_align", "k_hscan", "k_hctrl")): continue
        print(k, {a: f"{b:.4g}" for a, b in sorted(c.items())})
