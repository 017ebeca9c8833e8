"""Whole-batch comparison of the records the C ABI returns with the oracle's batch driver (tests only): every field of every
unit plus the four work counters that the roofline numerator is made of.  Returns {field: number of mismatching units}."""
import json
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def usable_cpus():
    import bench
    return bench.usable_cpus()


def _chk(bad, name, x, y):
    x, y = np.asarray(x), np.asarray(y)
    n = int((x != y).sum()) if x.shape == y.shape else -1
    if n:
        bad[name] = n


def compare_se(ores, hits, cc, nclass):
    bad = {}
    _chk(bad, "filtered", ores["filtered"] != 0, (hits["flags"] & 1) != 0)
    _chk(bad, "len", ores["len"], hits["len"])
    _chk(bad, "raw_len", ores["raw_len"], hits["raw_len"])
    ok = ores["filtered"] == 0
    _chk(bad, "max_snp", ores["read_max_snp_num"][ok], hits["max_snp"][ok])
    _chk(bad, "seedseg", ores["seedseg_num"][ok], hits["seedseg"][ok])
    _chk(bad, "n_hit", ores["n_hit"][ok][:, :nclass], cc["n_hit"][ok][:, :nclass])
    _chk(bad, "n_chit", ores["n_chit"][ok][:, :nclass], cc["n_chit"][ok][:, :nclass])
    _chk(bad, "n_best", np.maximum(ores["n_best"], 0)[ok], hits["n_best"][ok])
    has = ok & (ores["n_best"] > 0)
    for f in ("chr", "loc", "best_class"):
        _chk(bad, f, ores[f][has], hits[f][has])
    _chk(bad, "chain", ores["chain"][has] != 0, (hits["flags"][has] & 2) != 0)
    n_limit = int(((hits["flags"] & 4) != 0).sum())   # BSX_F_LIMIT: the one capacity the reference does not have (include/bsx.h) — never reached
    if n_limit:
        bad["BSX_F_LIMIT"] = n_limit
    return bad, {"placed": int(has.sum()), "filtered": int((~ok).sum()), "flagged_BSX_F_LIMIT": n_limit}


def compare_pe(ores, out, ca, cb, npairs, nclass):
    bad = {}
    _chk(bad, "paired", ores["paired"], out["paired"])
    both = (ores["a"]["filtered"] == 0) & (ores["b"]["filtered"] == 0)
    _chk(bad, "n_pairs", ores["n_pairs"][both][:, :2 * nclass - 1], npairs[both][:, :2 * nclass - 1])
    up = (ores["tmp"] == 1) | (ores["paired"] == 0)
    _chk(bad, "unpaired_out", up, out["unpaired_out"] != 0)
    pr = ~up
    for f in ("chain", "na", "nb", "insert", "a_chr", "a_loc", "b_chr", "b_loc"):
        _chk(bad, "pick." + f, ores["pick"][f][pr], out[f][pr])
    pd = ores["paired"] > 0
    _chk(bad, "pair_class", ores["pair_class"][pd], out["pair_class"][pd])
    _chk(bad, "pair_n", ores["pair_n"][pd], out["n_pairs"][pd])
    for m, cnts in (("a", ca), ("b", cb)):
        o, g = ores[m], out[m]
        _chk(bad, m + ".filtered", o["filtered"] != 0, (g["flags"] & 1) != 0)
        _chk(bad, m + ".len", o["len"], g["len"])
        _chk(bad, m + ".raw_len", o["raw_len"], g["raw_len"])
        ok = o["filtered"] == 0
        _chk(bad, m + ".max_snp", o["read_max_snp_num"][ok], g["max_snp"][ok])
        _chk(bad, m + ".seedseg", o["seedseg_num"][ok], g["seedseg"][ok])
        _chk(bad, m + ".n_hit", o["n_hit"][ok][:, :nclass], cnts["n_hit"][ok][:, :nclass])
        _chk(bad, m + ".n_chit", o["n_chit"][ok][:, :nclass], cnts["n_chit"][ok][:, :nclass])
        sel = up & ok & (o["n_best"] > 0)
        _chk(bad, m + ".n_best", o["n_best"][sel], g["n_best"][sel])
        for f in ("chr", "loc", "best_class"):
            _chk(bad, f"{m}.{f}", o[f][sel], g[f][sel])
        _chk(bad, m + ".chain", o["chain"][sel] != 0, (g["flags"][sel] & 2) != 0)
    return bad, {"paired_out": int(pr.sum()), "filtered_mates": int((ores["a"]["filtered"] != 0).sum() + (ores["b"]["filtered"] != 0).sum())}


def run_oracle(O, oref, al, pe, quals, K, leak_mode=0):
    """the oracle's batch driver over units [0, K) of device batch `al` on every CPU this process may use"""
    b1, o1 = al.download_reads(0)
    q1 = al.download_quals(0) if quals else None
    e1 = int(o1[K])
    t0 = time.time()
    if pe:
        b2, o2 = al.download_reads(1)
        q2 = al.download_quals(1) if quals else None
        e2 = int(o2[K])
        res, cnt = O.pe_batch(oref, b1[:e1], o1[:K + 1].copy(), b2[:e2], o2[:K + 1].copy(), q1[:e1] if quals else None, q2[:e2] if quals else None,
                              threads=usable_cpus(), leak_mode=leak_mode)
    else:
        res, cnt = O.se_batch(oref, b1[:e1], o1[:K + 1].copy(), q1[:e1] if quals else None, threads=usable_cpus(), leak_mode=leak_mode)
    return res, [int(x) for x in cnt], time.time() - t0


ROUND = "r05"


def record(name, info):
    """profiles/<round>_validate_<cfg>.json is a copy of what this writes on the GPU box (gpurun_out/ travels back)"""
    import bench
    d = os.environ.get("BSX_VALIDATE_DIR") or os.path.join(ROOT, "gpurun_out", "validate")
    os.makedirs(d, exist_ok=True)
    info = dict(info, config=name, lib_sha16=bench.lib_sha16(), oracle_threads=usable_cpus())
    with open(os.path.join(d, f"{ROUND}_validate_{name}.json"), "w") as f:
        json.dump(info, f, indent=1, sort_keys=True)
