#!/usr/bin/env python3
"""text Gantt of a command-line run made with BSX_TIMING=2 (tools/e2e_bench.py output): per batch, when it was parsed, uploaded, aligned,
read back, formatted and written, and what each stage waited for.  usage: e2e_gantt.py <e2e.json>"""
import json, sys
d = json.load(open(sys.argv[1]))
ev = d["timing"]["events"]
names = ["parse", "upload", "align", "readback", "format", "write"]
by = {}
for k, st, a, b in ev:
    by.setdefault(k, {})[st] = (a, b)
print("batch " + "  ".join("%-15s" % n for n in names))
for k in sorted(by):
    print("%5d " % k + "  ".join(("%6.3f-%6.3f " % by[k][s]) if s in by[k] else " " * 15 for s in range(6)))
# gaps between consecutive stages of the same batch (time a finished batch waited for the next stage)
print("waits (s) between stages, per batch:")
for k in sorted(by):
    e = by[k]
    w = []
    for s0, s1 in ((0, 1), (3, 4), (4, 5)):
        if s0 in e and s1 in e:
            w.append("%s->%s %.3f" % (names[s0], names[s1], e[s1][0] - e[s0][1]))
    print("%5d " % k + "  ".join(w))
