#!/usr/bin/env python3
"""Golden for `-o x.bam` (reference main.cpp:466-473 + sam2bam.sh): the SAM text the REAL bsmap binary wrote (kept in
cli_outputs.json.gz) taken through the reference's own vendored samtools 0.1.7a (oracle/_ref/samtools, built by
`make -C oracle samtools`) exactly as sam2bam.sh does — view -bS, sort, index — then decoded: header, records in sorted
order, and the index with its virtual offsets translated to record ordinals.  Run in the build container only."""
import gzip
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bam_util  # noqa: E402

SAMTOOLS = os.path.join(ROOT, "oracle", "_ref", "samtools")


def main():
    cli = json.load(gzip.open(os.path.join(HERE, "cli_outputs.json.gz"), "rt"))
    out = {}
    for name in sorted(cli):
        for tag in ("sam_Ru", "sam_plain"):
            run = cli[name][tag]
            if not run["out"]:
                continue
            with tempfile.TemporaryDirectory() as d:
                sam = os.path.join(d, "x.bam")       # the reference writes its SAM text under the .bam name first
                open(sam, "w").write(run["out"])
                tmp = os.path.join(d, "x.tmp.bam")
                with open(tmp, "wb") as f:
                    subprocess.check_call([SAMTOOLS, "view", "-bS", sam], stdout=f, stderr=subprocess.DEVNULL)
                subprocess.check_call([SAMTOOLS, "sort", tmp, os.path.join(d, "x")], stderr=subprocess.DEVNULL)
                subprocess.check_call([SAMTOOLS, "index", os.path.join(d, "x.bam")], stderr=subprocess.DEVNULL)
                bam = bam_util.decode_bam(os.path.join(d, "x.bam"))
                bai = bam_util.decode_bai(os.path.join(d, "x.bam.bai"), bam)
            out[f"{name}/{tag}"] = dict(options=run["options"], refs=bam["refs"], header_text=bam["header_text"], records=bam["records"],
                                        index=[dict(bins={str(b): c for b, c in bins.items()}, linear=lin) for bins, lin in bai])
            print(name, tag, len(bam["records"]), "records", sum(len(b) for b, _ in bai), "bins")
    with gzip.open(os.path.join(HERE, "cli_bamout.json.gz"), "wt") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
