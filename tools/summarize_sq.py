#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc SQ/TA/TCC counter passes per kernel into profiles/<tag>_sq.json.

usage: summarize_sq.py <tag> <pass_dir> [<pass_dir> ...]        (each dir = one `rocprofv3 --pmc ... --kernel-trace -d <dir>` run,
                                                                 CSV (`--output-format csv`) or the default rocpd .db)
Derived per kernel (units per MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles;
GRBM_GUI_ACTIVE is summed over the 8 XCDs):
  clock_GHz            = GRBM_GUI_ACTIVE / 8 / kernel time
  valu_instr_per_s     = SQ_INSTS_VALU / kernel time
  valu_busy_frac_4cyc  = SQ_INSTS_VALU * 4 / (n_simd * GRBM_GUI_ACTIVE / 8)   (issue cost measured by tools/microbench/valu_issue)
  valu_busy_frac_2cyc  = the same at 2 cycles per wave64 instruction
  wait_inst_frac       = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES,   wait_any_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES
  ta_busy_frac         = TA_TA_BUSY_sum / (n_cu * GRBM_GUI_ACTIVE / 8)
"""
import collections
import csv
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUTDIR = os.environ.get("BSX_PROFILES_DIR") or os.path.join(ROOT, "profiles")  # (on the GPU box: a directory under gpurun_out/)
KERNELS = ("k_align", "k_hscan", "k_hctrl", "k_plan", "k_scan")
N_CU, N_SIMD = 256, 1024


def short(name):
    for k in KERNELS:
        if k in name:
            if k == "k_hscan":   # (the scan kernels' one template argument is the work-counter switch: a serial-mode run holds one of the two, named by the file's work_counters)
                return k
            t = "<true>" if "<true>" in name else "<false>" if "<false>" in name else ""
            return k + t
    return None


def read_pass(d):
    """-> {kernel: {counter: sum}}, {kernel: (launches, total_ns)}"""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(dict)
    csvs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if csvs:
        for f in csvs:
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if not k:
                    continue
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    else:
        for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
            c = sqlite3.connect(f)
            for name, ctr, val, did, s, e in c.execute("select kernel_name, counter_name, value, dispatch_id, start, end from counters_collection"):
                k = short(name)
                if not k:
                    continue
                acc[k][ctr] += float(val)
                disp[k][did] = int(e) - int(s)
    return acc, {k: (len(v), sum(v.values())) for k, v in disp.items()}


def main():
    tag, dirs = sys.argv[1], sys.argv[2:]
    out = {"tag": tag, "passes": [], "kernels": {}}
    try:   # which build the counters belong to (bench.py only quotes a summary of the library it runs)
        sys.path.insert(0, ROOT)
        import bench
        out["lib_sha16"] = bench.lib_sha16()
    except Exception:
        pass
    # what was profiled (set by the profiling scripts): bench.py only quotes a summary for the mode and counter setting it runs
    out["mode"] = os.environ.get("BSX_PROFILE_MODE", "pe")
    out["work_counters"] = int(os.environ.get("BSX_PROFILE_WORK_COUNTERS", "0"))
    out["steps_in_pass"] = int(os.environ.get("BSX_PROFILE_STEPS", "3"))          # warm-up + timed steps of each counter pass: counters are sums over all of them
    try:   # units per step of the profiled run: given, or the mode's default step size (what `bench.py --profile-serial` runs without --pairs-per-step)
        out["units_per_step"] = int(os.environ["BSX_PROFILE_UNITS"]) if os.environ.get("BSX_PROFILE_UNITS") else bench.mode_defaults(out["mode"])[0]
    except Exception:
        out["units_per_step"] = None
    out["scan_kernel"] = os.environ.get("BSX_SAME", "1")
    merged = collections.defaultdict(dict)
    for d in dirs:
        acc, disp = read_pass(d)
        out["passes"].append({"dir": os.path.basename(d.rstrip("/")), "counters": sorted({c for v in acc.values() for c in v})})
        for k, ctrs in acc.items():
            n, ns = disp[k]
            for c, v in ctrs.items():
                merged[k][c] = {"sum": v, "launches": n, "kernel_ms": ns / 1e6}
    for k, ctrs in sorted(merged.items()):
        e = {"counters": {c: v["sum"] for c, v in sorted(ctrs.items())}}
        g = lambda c: ctrs[c]["sum"] if c in ctrs else None
        t = lambda c: ctrs[c]["kernel_ms"] * 1e-3 if c in ctrs else None
        e["launches"] = max(v["launches"] for v in ctrs.values())
        e["kernel_ms_total"] = {c: round(v["kernel_ms"], 3) for c, v in ctrs.items() if c in ("SQ_INSTS_VALU", "GRBM_GUI_ACTIVE", "TA_TA_BUSY_sum", "SQ_WAIT_INST_ANY")}
        d = {}
        if g("GRBM_GUI_ACTIVE"):
            d["clock_GHz"] = g("GRBM_GUI_ACTIVE") / 8 / t("GRBM_GUI_ACTIVE") / 1e9
        clk = d.get("clock_GHz", 2.2) * 1e9
        if g("SQ_INSTS_VALU"):
            rate = g("SQ_INSTS_VALU") / t("SQ_INSTS_VALU")
            d["valu_instr_per_s"] = rate
            d["valu_busy_frac_4cyc"] = rate * 4 / (N_SIMD * clk)
            d["valu_busy_frac_2cyc"] = rate * 2 / (N_SIMD * clk)
        for c, name in (("SQ_INSTS_SALU", "salu_per_valu"), ("SQ_INSTS_VMEM_RD", "vmem_rd_per_valu"), ("SQ_INSTS_LDS", "lds_per_valu")):
            if g(c) is not None and g("SQ_INSTS_VALU"):
                d[name] = g(c) / g("SQ_INSTS_VALU")
        if g("SQ_WAVE_CYCLES"):
            for c, name in (("SQ_WAIT_INST_ANY", "wait_inst_frac"), ("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_ACTIVE_INST_VALU", "active_valu_frac")):
                if g(c) is not None:
                    d[name] = g(c) / g("SQ_WAVE_CYCLES")  # (different passes: same kernel, same work)
            if g("SQ_WAVES"):
                d["quad_cycles_per_wave"] = g("SQ_WAVE_CYCLES") / g("SQ_WAVES")
        if g("SQ_WAVE_CYCLES") and g("GRBM_GUI_ACTIVE"):   # quad-cycles of resident waves over the SIMDs' cycles (different passes: same kernel, same work)
            d["resident_waves_per_simd"] = g("SQ_WAVE_CYCLES") * 4 / (N_SIMD * g("GRBM_GUI_ACTIVE") / 8)
        if g("TCC_EA0_RDREQ_sum") is not None:
            d["fabric_read_requests_per_step"] = g("TCC_EA0_RDREQ_sum") / out["steps_in_pass"]
            d["fabric_read_requests_per_s"] = g("TCC_EA0_RDREQ_sum") / t("TCC_EA0_RDREQ_sum")
        if g("TA_TA_BUSY_sum"):
            d["ta_busy_frac"] = g("TA_TA_BUSY_sum") / (N_CU * clk * t("TA_TA_BUSY_sum"))
        if g("TCP_TOTAL_CACHE_ACCESSES_sum") and g("TCP_TCC_READ_REQ_sum") is not None:
            d["l1_miss_per_access"] = g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum")
        if g("TA_TA_BUSY_sum") and g("SQ_INSTS_VMEM_RD"):
            d["ta_busy_cycles_per_vmem_instr"] = g("TA_TA_BUSY_sum") / g("SQ_INSTS_VMEM_RD")
        if g("SQ_LDS_IDX_ACTIVE") is not None and g("GRBM_GUI_ACTIVE"):   # LDS-array cycles over the CUs' cycles (counter passes differ: same kernel, same work)
            d["lds_active_frac"] = g("SQ_LDS_IDX_ACTIVE") / (N_CU * g("GRBM_GUI_ACTIVE") / 8)
            if g("SQ_LDS_BANK_CONFLICT") is not None:
                d["lds_conflict_frac_of_active"] = g("SQ_LDS_BANK_CONFLICT") / max(1.0, g("SQ_LDS_IDX_ACTIVE"))
        if g("SQ_WAIT_INST_LDS") is not None and g("SQ_WAVE_CYCLES"):
            d["wait_inst_lds_frac"] = g("SQ_WAIT_INST_LDS") / g("SQ_WAVE_CYCLES")
        if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
            d["l2_hit_frac"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
        e["derived"] = d
        out["kernels"][k] = e
    os.makedirs(OUTDIR, exist_ok=True)
    json.dump(out, open(os.path.join(OUTDIR, f"{tag}_sq.json"), "w"), indent=1)
    print(json.dumps({k: v["derived"] for k, v in out["kernels"].items()}, indent=1))


if __name__ == "__main__":
    main()
