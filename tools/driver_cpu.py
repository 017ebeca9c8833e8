#!/usr/bin/env python3
"""GPU box: how much CPU does the host thread that drives a batch burn?  One C3 batch of 2^20 pairs on the hg38-sized genome, Do_Batch
timed with the wall clock and with the calling thread's CPU clock.  usage: driver_cpu.py [--steps 4]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (first: libbsx.so binds to the HIP runtime torch has loaded)
import bench as BN
import bsmap_amd as B

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=4); a = ap.parse_args()
M = BN.MODES["pe"]
ref = B.RefSeq(B.make_params(**M["kw"])).synthetic(BN.HG38, seed=38).CreateIndex()
n = 1 << 20
al = B.PairAlign(ref, n * (a.steps + 1))
al.synth_reads(n * (a.steps + 1), M["L"], seed=3, kind=M["kind"])
al.run_range(0, n, sync=True)
out = []
for i in range(1, a.steps + 1):
    w0, c0 = time.perf_counter(), time.thread_time()
    al.run_range(i * n, n, sync=True)
    out.append({"wall_ms": (time.perf_counter() - w0) * 1e3, "thread_cpu_ms": (time.thread_time() - c0) * 1e3})
print(json.dumps({"do_batch": out, "env": {k: v for k, v in os.environ.items() if k.startswith(("BSX_", "HIP_", "HSA_", "GPU_", "AMD_"))}}))
al.close(); ref.close()
