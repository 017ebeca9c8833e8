"""Loader for the committed golden vectors (tests/golden/, produced by make_golden.py from the real reference)."""
import gzip
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CONFIGS = ["c1_se36", "c2_se100", "c2_se100_n1", "c2_se100_r0_w3", "c3_pe150", "c3_pe150_r0", "c4_rrbs75", "c5_trim_pe150"]


def load(name):
    meta = json.load(gzip.open(os.path.join(GOLDEN, name + ".json.gz"), "rt"))
    arr = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    rrbs = "D" in meta["kw"]
    fasta = os.path.join(GOLDEN, "genome_rrbs.fa" if rrbs else "genome_wgbs.fa")
    return meta, arr, fasta


def sparse_index(bucket_off, bucket_nfwd=None):
    cnt = np.diff(bucket_off.astype(np.int64))
    keys = np.nonzero(cnt)[0].astype(np.uint32)
    return keys, cnt[keys].astype(np.uint32), (bucket_nfwd[keys] if bucket_nfwd is not None else None)
