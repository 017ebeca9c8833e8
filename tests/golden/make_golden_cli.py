#!/usr/bin/env python3
"""Golden outputs of the REAL `bsmap` binary (oracle/_ref/bsmap, built from /root/reference) for the command-line
driver tests: SAM with -R -u, BSP with -u (+ the -2 file for pairs), on the reads of the existing golden sets.
Run in the build container only.  Stores command-line options and output text; no reference source."""
import gzip
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as G  # noqa: E402
from oracle import ref_ffi as R  # noqa: E402


def opts_from_kw(kw):
    o = []
    if "D" in kw:
        o += ["-D", kw["D"]]
    for k in ("s", "v", "I", "S", "r", "n", "w", "m", "x", "q", "f", "z", "L", "M"):
        if kw.get(k) is not None:
            o += ["-" + k, str(kw[k])]
    for a in kw.get("A") or []:
        o += ["-A", a]
    return o


def main():
    assert R.build()
    tmp = tempfile.mkdtemp()
    out = {}
    for name in ("c1_se36", "c2_se100_n1", "c2_se100_r0_w3", "c3_pe150", "c3_pe150_r0", "c4_rrbs75", "c5_trim_pe150"):
        meta, arr, fasta = G.load(name)
        kw = meta["kw"]
        pe = meta["kind"] == "pe"
        if pe:
            f1, f2 = os.path.join(tmp, name + "_1.fq"), os.path.join(tmp, name + "_2.fq")
            with open(f1, "w") as a, open(f2, "w") as b:
                for r in meta["reads"]:
                    a.write(f"@{r['name']}/1\n{r['seq1']}\n+\n{r['qual1']}\n")
                    b.write(f"@{r['name']}/2\n{r['seq2']}\n+\n{r['qual2']}\n")
            inputs = ["-a", f1, "-b", f2]
        else:
            f1 = os.path.join(tmp, name + ".fq")
            with open(f1, "w") as a:
                for r in meta["reads"]:
                    a.write(f"@{r['name']}\n{r['seq']}\n+\n{r['qual']}\n")
            inputs = ["-a", f1]
        runs = {}
        for tag, extra, ext in (("sam_Ru", ["-R", "-u"], ".sam"), ("bsp_u", ["-u"], ".bsp"), ("sam_plain", [], ".sam")):
            o = os.path.join(tmp, f"{name}_{tag}{ext}")
            o2 = os.path.join(tmp, f"{name}_{tag}_unpair{ext}")
            args = opts_from_kw(kw) + extra
            cmd = inputs + ["-d", fasta, "-o", o, "-p", "1"] + (["-2", o2] if pe and ext == ".bsp" else []) + args
            # -D must come first on the real command line (main.cpp:247,257)
            if "D" in kw:
                cmd = ["-D", kw["D"]] + [x for i, x in enumerate(cmd) if not (x == "-D" or (i > 0 and cmd[i - 1] == "-D"))]
            R.run_bsmap(cmd)
            runs[tag] = dict(options=args, ext=ext, out=open(o).read(), out2=open(o2).read() if os.path.exists(o2) else None)
        out[name] = runs
        print(name, {k: v["out"].count("\n") for k, v in runs.items()})
    json.dump(out, gzip.open(os.path.join(HERE, "cli_outputs.json.gz"), "wt"))


if __name__ == "__main__":
    main()
