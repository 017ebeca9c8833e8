#!/usr/bin/env python3
"""Byte-level golden for `-o x.bam` on a file of several BGZF blocks: the SAM body of the c3_pe150 golden repeated twelve times
(names made distinct) through the reference's vendored samtools 0.1.7a as sam2bam.sh runs it (view -bS | sort | index); stored:
the SHA-256 and size of x.bam and x.bam.bai.  Run in the build container only."""
import gzip, hashlib, json, os, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SAMTOOLS = os.path.join(ROOT, "oracle", "_ref", "samtools")


def big_sam():
    cli = json.load(gzip.open(os.path.join(HERE, "cli_outputs.json.gz"), "rt"))
    txt = cli["c3_pe150"]["sam_Ru"]["out"]
    hdr = [l for l in txt.split("\n") if l.startswith("@")]
    body = [l for l in txt.split("\n") if l and not l.startswith("@")]
    big = []
    for k in range(12):
        for l in body:
            f = l.split("\t")
            f[0] = f"{f[0]}_{k}"
            big.append("\t".join(f))
    return "\n".join(hdr + big) + "\n"


def main():
    d = tempfile.mkdtemp()
    sam = os.path.join(d, "in.sam")
    open(sam, "w").write(big_sam())
    with open(os.path.join(d, "t.bam"), "wb") as f:
        subprocess.check_call([SAMTOOLS, "view", "-bS", sam], stdout=f, stderr=subprocess.DEVNULL)
    subprocess.check_call([SAMTOOLS, "sort", os.path.join(d, "t.bam"), os.path.join(d, "x")], stderr=subprocess.DEVNULL)
    subprocess.check_call([SAMTOOLS, "index", os.path.join(d, "x.bam")], stderr=subprocess.DEVNULL)
    out = {}
    for n in ("x.bam", "x.bam.bai"):
        b = open(os.path.join(d, n), "rb").read()
        out[n] = {"sha256": hashlib.sha256(b).hexdigest(), "bytes": len(b)}
    json.dump(out, open(os.path.join(HERE, "bam_multiblock.json"), "w"), indent=1)
    print(out)


if __name__ == "__main__":
    main()
