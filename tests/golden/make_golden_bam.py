#!/usr/bin/env python3
"""Golden outputs of the REAL `bsmap` binary (oracle/_ref/bsmap) on BAM input (reads.cpp:38-41,120-142): the reads of two
golden sets written as unaligned BAM (tests/bam_util.py), single-end (-a x.bam) and paired (-a x.bam -b x.bam, mates in
alternating records).  Run in the build container only.  Stores options and output text."""
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
import bam_util  # noqa: E402


def bam_records(meta):
    if meta["kind"] == "pe":
        recs = []
        for r in meta["reads"]:
            recs.append(bam_util.record(r["name"], r["seq1"], r["qual1"], flag=0x4D))
            recs.append(bam_util.record(r["name"], r["seq2"], r["qual2"], flag=0x8D))
        return recs
    return [bam_util.record(r["name"], r["seq"], r["qual"], flag=4) for r in meta["reads"]]


def main():
    from oracle import ref_ffi as R
    from make_golden_cli import opts_from_kw
    assert R.build()
    tmp = tempfile.mkdtemp()
    out = {}
    for name in ("c2_se100_n1", "c3_pe150", "c5_trim_pe150"):
        meta, arr, fasta = G.load(name)
        kw = meta["kw"]
        pe = meta["kind"] == "pe"
        bam = os.path.join(tmp, name + ".bam")
        bam_util.write_bam(bam, bam_records(meta), block=7000)
        inputs = ["-a", bam] + (["-b", bam] if pe else [])
        runs = {}
        for tag, extra in (("sam_Ru", ["-R", "-u"]), ("sam_B3", ["-u", "-B", "3", "-E", "40"])):
            o = os.path.join(tmp, f"{name}_{tag}.sam")
            args = opts_from_kw(kw) + extra
            R.run_bsmap(inputs + ["-d", fasta, "-o", o, "-p", "1"] + args)
            runs[tag] = dict(options=args, out=open(o).read())
        out[name] = runs
        print(name, {k: v["out"].count("\n") for k, v in runs.items()})
    json.dump(out, gzip.open(os.path.join(HERE, "cli_bam_outputs.json.gz"), "wt"))


if __name__ == "__main__":
    main()
