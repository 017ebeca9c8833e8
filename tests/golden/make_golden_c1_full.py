#!/usr/bin/env python3
"""BASELINE.json's configs[0] at its stated size — 10 000 single-end 36 bp reads against a 1 Mb reference, `-s 12 -v 2 -p 1` —
through the REAL `bsmap` binary (oracle/_ref/bsmap, built from /root/reference).  The inputs come from the seeded generators
of tests/bsx_testdata.py (our code), so only their digests and the binary's SAM text are stored (c1_full.json.gz).
Run in the build container only."""
import gzip
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bsx_testdata as td  # noqa: E402
from oracle import ref_ffi as R  # noqa: E402

OPTS = ["-s", "12", "-v", "2", "-S", "1"]


def main():
    assert R.build(), "needs /root/reference"
    tmp = tempfile.mkdtemp()
    fa, fq, h = td.c1_full_inputs(tmp)
    out = os.path.join(tmp, "c1.sam")
    R.run_bsmap(["-a", fq, "-d", fa, "-o", out, "-p", "1"] + OPTS)
    sam = open(out).read()
    json.dump(dict(options=OPTS, inputs_sha256=h, sam=sam), gzip.open(os.path.join(HERE, "c1_full.json.gz"), "wt"))
    print("reads 10000, sam lines", sam.count("\n"), "inputs", h[:16])


if __name__ == "__main__":
    main()
