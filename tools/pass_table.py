#!/usr/bin/env python3
"""per-pass table of the heavy pipeline from a rocprofv3 --kernel-trace csv of a SERIAL-mode bench run (one batch in flight, one unit group):
for the last whole step, every control pass with its k_hctrl, order-kernel and scan durations (us) — where the control time of a step goes by pass index.
usage: pass_table.py <trace dir> [bench stderr with BSX_TRACE_HEAVY lines]"""
import csv, glob, json, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
starts = [i for i, r in enumerate(rows) if "k_align" in r[2]]
a, b = (starts[-2], starts[-1]) if len(starts) > 1 else (starts[-1], len(rows))
passes = []; cur = None
for s, e, k in rows[a:b]:
    d = (e - s) / 1e3
    if "k_hctrl" in k:
        cur = {"t_us": round((s - rows[a][0]) / 1e3), "ctrl": round(d, 1), "order": 0.0, "scan": 0.0}; passes.append(cur)
    elif cur is None: continue
    elif "k_hscan" in k: cur["scan"] = round(cur["scan"] + d, 1)
    elif any(x in k for x in ("k_task", "k_bin")): cur["order"] = round(cur["order"] + d, 1)
act = []
if len(sys.argv) > 2:
    for ln in open(sys.argv[2]):
        m = re.search(r"base (\d+) group \d+ passes (\d+) active (\d+) tasks (\d+)", ln)
        if m: act.append(tuple(int(x) for x in m.groups()))
tot = {k: round(sum(p[k] for p in passes) / 1e3, 2) for k in ("ctrl", "order", "scan")}
out = {"step_ms": round((rows[b][0] - rows[a][0]) / 1e6, 2) if b < len(rows) else None, "k_align_ms": round((rows[a][1] - rows[a][0]) / 1e6, 2), "passes": len(passes), "ms": tot,
       "ctrl_ms_by_decile_of_passes": [round(sum(p["ctrl"] for p in passes[i * len(passes) // 10:(i + 1) * len(passes) // 10]) / 1e3, 2) for i in range(10)],
       "scan_ms_by_decile_of_passes": [round(sum(p["scan"] for p in passes[i * len(passes) // 10:(i + 1) * len(passes) // 10]) / 1e3, 2) for i in range(10)]}
print(json.dumps(out))
print("# pass  t_us  ctrl_us  order_us  scan_us")
for i, p in enumerate(passes): print(i, p["t_us"], p["ctrl"], p["order"], p["scan"])
if act:
    print("# host polls (every chunk of passes, last Do_Batch calls): round-base passes active tasks")
    for x in act[-120:]: print(*x)
