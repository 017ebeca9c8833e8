#!/usr/bin/env python3
"""per-step timeline summary of a rocprofv3 --kernel-trace run of bench.py: busy time (union of kernel intervals), idle gaps,
launches per step and the control passes' lengths.  usage: timeline.py <trace dir> [steps]  -> one JSON object"""
import csv, glob, json, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
def short(k):
    for s in ("k_hscan_shared", "k_hscan", "k_hctrl", "k_align", "k_leak", "k_task_keys", "k_hsort", "k_synth"):
        if s in k: return s
    return "sort(rocprim)" if "rocprim" in k or "radix" in k.lower() or "onesweep" in k.lower() else "other"
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows)
starts = [i for i, e in enumerate(ev) if e[2] == "k_align"]
steps = []
for a, b in zip(starts, starts[1:] + [len(ev)]):
    seg = [e for e in ev[a:b] if e[2] != "k_synth"]
    if len(seg) < 3: continue   # (the redo run of k_align follows its step directly)
    t0, t1 = seg[0][0], max(e[1] for e in seg)
    busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
    for s, e, _ in seg[1:]:
        if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    per = {}
    for s, e, k in seg:
        d = per.setdefault(k, [0, 0.0, 0.0]); d[0] += 1; d[1] += (e - s) / 1e6; d[2] = max(d[2], (e - s) / 1e6)
    steps.append({"span_ms": (t1 - t0) / 1e6, "busy_ms": busy / 1e6, "idle_ms": (t1 - t0 - busy) / 1e6, "launches": len(seg),
                  "kernels": {k: {"n": v[0], "ms": round(v[1], 3), "max_ms": round(v[2], 3)} for k, v in per.items()}})
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 1
use = steps[skip:] or steps
agg = {"steps_seen": len(steps), "span_ms": sum(s["span_ms"] for s in use) / len(use), "busy_ms": sum(s["busy_ms"] for s in use) / len(use),
       "idle_ms": sum(s["idle_ms"] for s in use) / len(use), "launches": sum(s["launches"] for s in use) / len(use), "last_step": use[-1]}
print(json.dumps(agg))
