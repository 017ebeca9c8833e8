#!/usr/bin/env python3
"""Random 16-byte gather rate of the device as a function of the window it gathers from (L2 4 MB/XCD, Infinity Cache 256 MB,
HBM beyond) — the ceiling the planner's bucket-size gathers and the scan's reference gathers run against.  GPU box only."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (HIP runtime first)
import bsmap_amd as B
out = {}
for mb in (2, 16, 64, 128, 192, 512, 1024, 4096):
    r = B.probe_memory(0, 4 << 30, mb << 20)
    out[f"{mb}MB"] = {"gather16_Gloads_per_s": round(r["gather16_Gloads_per_s"], 2), "stream_read_GBps": round(r["stream_read"], 1), "stream_copy_GBps": round(r["stream_copy"], 1)}
print(json.dumps(out, indent=1))
