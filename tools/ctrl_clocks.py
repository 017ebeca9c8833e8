#!/usr/bin/env python3
"""category clocks of the heavy control kernel on a bench-sized step (diagnostic, GPU box).  usage: ctrl_clocks.py --mode pe|se|rrbs|trim [--units N]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench as BN
import bsmap_amd as B
ap = argparse.ArgumentParser(); ap.add_argument("--mode", default="pe"); ap.add_argument("--units", type=int, default=1 << 20); ap.add_argument("--genome", default="hg38"); ap.add_argument("--work-counters", type=int, default=0, help="0 (default): as the command line and the bench's timed region run")
a = ap.parse_args()
M = BN.MODES[a.mode]
lens = BN.HG38 if a.genome == "hg38" else [max(200000, int(x * float(a.genome))) for x in BN.HG38]
ref = B.RefSeq(B.make_params(**M["kw"])).synthetic(lens, seed=38).CreateIndex()
bt = (B.PairAlign if M["pe"] else B.SingleAlign)(ref, a.units).set_work_counters(bool(a.work_counters))
bt.synth_reads(a.units, M["L"], seed=3, kind=M["kind"])
bt.Do_Batch(); bt.Do_Batch(); ms = bt.kernel_ms()
bt.set_debug(2); bt.reset_counters(); bt.Do_Batch()
cc = bt.ctrl_clocks().astype(float)
names = ["prepare/restore", "inline-scan", "replay", "sort+pairs", "save/finish", "recount", "advance-total", "overflow-rest"]
out = {"mode": a.mode, "units": a.units, "work_counters": bool(a.work_counters), "kernel_ms": ms, "kernel_ms_clocks_on": bt.kernel_ms(), "heavy_units": bt.heavy_units(), "redo_units": bt.redo_units(),
       "sum_Mcycles": dict(zip(names, (cc[:8] / 1e6).round(1).tolist())), "longest_span_kcycles": dict(zip(names, (cc[8:16] / 1e3).round(1).tolist())),
       "longest_visit_breakdown_kcycles_x_spans": {n: [round((int(v) >> 16) / 1e3, 1), int(v) & 0xffff] for n, v in zip(["restore", "inline-scan", "replay", "sort+pairs", "save", "recount"], cc[16:22])}}
c = bt.unit_cycles().astype(np.float64)
out["main_kernel_unit_cycles"] = {"mean": c.mean(), "p50": float(np.median(c)), "p99": float(np.percentile(c, 99)), "max": c.max()}
print(json.dumps(out))
