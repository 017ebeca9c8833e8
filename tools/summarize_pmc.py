#!/usr/bin/env python3
"""Summarise rocprofv3 outputs into profiles/: per-kernel stats (from --kernel-trace --stats) and HBM traffic per launch
from the FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, as MI355X_MICROARCH.md prescribes).

gfx950 correction applied (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports half of the bytes of wide coalesced reads, so
the read side is given both raw and doubled; rocprofv3 reports both counters in KiB.
usage: summarize_pmc.py <tag> <stats_dir> <fetch_dir> <write_dir>"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUTDIR = os.environ.get("BSX_PROFILES_DIR") or os.path.join(ROOT, "profiles")  # (on the GPU box: a directory under gpurun_out/)


def per_kernel(dirname, counter):
    out = {}
    for f in glob.glob(os.path.join(dirname, "**", "*_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            k = row["Kernel_Name"]
            d = out.setdefault(k, {"launches": 0, "sum": 0.0, "ns": 0})
            d["launches"] += 1
            d["sum"] += float(row["Counter_Value"])
            d["ns"] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    return out


def main():
    tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
    os.makedirs(OUTDIR, exist_ok=True)
    for f in glob.glob(os.path.join(stats_dir, "**", "*_kernel_stats.csv"), recursive=True):
        rows = list(csv.reader(open(f)))
        with open(os.path.join(OUTDIR, f"{tag}_kernel_stats.csv"), "w") as o:
            w = csv.writer(o)
            for r in rows:
                r[0] = r[0][:120]
                w.writerow(r)
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    summ = {}
    for k in sorted(set(fetch) | set(write)):
        if not any(t in k for t in ("k_align", "k_hscan", "k_hctrl")):
            continue
        f, w = fetch.get(k), write.get(k)
        e = {"launches": (f or w)["launches"]}
        if f:
            e["FETCH_SIZE_KiB_per_launch"] = f["sum"] / f["launches"]
            e["fetch_bytes_per_launch_raw"] = f["sum"] / f["launches"] * 1024
            e["fetch_bytes_per_launch_x2_gfx950"] = 2 * f["sum"] / f["launches"] * 1024
            e["avg_ms_in_pmc_pass"] = f["ns"] / f["launches"] / 1e6
        if w:
            e["WRITE_SIZE_KiB_per_launch"] = w["sum"] / w["launches"]
            e["write_bytes_per_launch"] = w["sum"] / w["launches"] * 1024
        summ[k[:100]] = e
    # one bench step (= one Do_Batch) is one k_align launch plus all heavy-pipeline iterations that follow it
    def steps(d):
        return max([v["launches"] for k, v in d.items() if "k_align" in k] or [1])
    fsteps, wsteps = steps(fetch), steps(write)
    fetch_step = sum(v["sum"] for k, v in fetch.items() if k[:100] in summ) * 1024 / fsteps
    write_step = sum(v["sum"] for k, v in write.items() if k[:100] in summ) * 1024 / wsteps
    for k, e in summ.items():
        e["launches_per_step"] = e["launches"] / fsteps
    import hashlib
    try:
        sha = hashlib.sha256(open(os.path.join(ROOT, "bsmap_amd", "libbsx.so"), "rb").read()).hexdigest()[:16]
    except OSError:
        sha = None
    out = {"tag": tag, "lib_sha16": sha, "steps_in_fetch_pass": fsteps, "steps_in_write_pass": wsteps, "kernels": summ,
           "fetch_bytes_per_step_raw": fetch_step, "fetch_bytes_per_step_x2_gfx950": 2 * fetch_step, "write_bytes_per_step": write_step,
           "hbm_bytes_per_launch": 2 * fetch_step + write_step,
           "note": "per bench step (one Do_Batch = k_align + heavy-pipeline iterations); read side = 2 x FETCH_SIZE "
                   "(gfx950 correction of MI355X_MICROARCH.md, uncalibrated for narrow random loads), write side = WRITE_SIZE"}
    json.dump(out, open(os.path.join(OUTDIR, f"{tag}_pmc.json"), "w"), indent=1)
    json.dump(out, open(os.path.join(OUTDIR, "pmc_latest.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
