#!/usr/bin/env python3
"""Throughput of the methylation pile-up kernels (bsx_meth_add) on an hg38-sized random reference: N alignments of L nt
at uniform positions, all four strands, reads = reference letters with bisulfite-like conversion.  Prints one JSON line;
run under `rocprofv3 --kernel-trace --stats` for the kernel-only times.  usage: meth_bench.py [--aln 16777216] [--len 100]"""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bsmap_amd as B
from bsmap_amd import methratio as MR


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--aln", type=int, default=16 << 20)
    ap.add_argument("--len", type=int, default=100)
    ap.add_argument("--genome", type=float, default=1.0)
    a = ap.parse_args()
    L = MR._bind()
    rng = np.random.default_rng(1)
    lens = np.array([int(x * a.genome) for x in (248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
                                                135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
                                                46709983, 50818468, 156040895, 57227415)], np.uint64)
    h = C.c_void_p()
    B._check(L.bsx_meth_create(len(lens), lens.ctypes.data, 1, 0, C.byref(h)))
    letters = np.frombuffer(b"ACGT", np.uint8)
    chunks = []
    for c, n in enumerate(lens.tolist()):
        s = letters[rng.integers(0, 4, n, dtype=np.uint8)]
        B._check(L.bsx_meth_set_reference(h, c, s.tobytes()))
        if c < 2:
            chunks.append(s)
    n, ln = a.aln, a.len
    # alignments on the first two chromosomes only need their letters on the host; positions are uniform there
    chr_ = rng.integers(0, 2, n).astype(np.uint32)
    pos = (rng.random(n) * (np.array([len(chunks[0]), len(chunks[1])])[chr_] - ln - 1)).astype(np.int64)
    strand = rng.integers(0, 4, n).astype(np.uint8)
    seq = np.empty((n, ln), np.uint8)
    for c in (0, 1):
        sel = np.nonzero(chr_ == c)[0]
        idx = pos[sel, None] + np.arange(ln)[None, :]
        seq[sel] = chunks[c][idx]
    conv = rng.random((n, ln)) < 0.7  # unmethylated fraction
    plus = (strand & 1) == 0
    seq[plus[:, None] & (seq == ord("C")) & conv] = ord("T")
    seq[(~plus)[:, None] & (seq == ord("G")) & conv] = ord("A")
    ins = np.zeros(n, np.int32); cut = np.full(n, -1, np.int64)
    off = (np.arange(n + 1, dtype=np.uint64) * ln)
    B._check(L.bsx_meth_add(h, min(n, 1 << 20), chr_.ctypes.data, pos.ctypes.data, strand.ctypes.data, ins.ctypes.data, cut.ctypes.data, seq.ctypes.data, off.ctypes.data, 2))
    t0 = time.perf_counter()
    B._check(L.bsx_meth_add(h, n, chr_.ctypes.data, pos.ctypes.data, strand.ctypes.data, ins.ctypes.data, cut.ctypes.data, seq.ctypes.data, off.ctypes.data, 2))
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    rows = cov = 0
    for c in range(len(lens)):
        nr, cv, sd = C.c_uint32(), C.c_uint64(), C.c_uint64()
        B._check(L.bsx_meth_report_chr(h, c, 1, 1, C.byref(nr), C.byref(cv), C.byref(sd)))
        rows += nr.value; cov += cv.value
    dt_rep = time.perf_counter() - t1
    L.bsx_meth_destroy(h)
    print(json.dumps({"alignments": n, "read_len": ln, "add_s_incl_pcie": round(dt, 3), "alignments_per_s_incl_pcie": round(n / dt),
                      "report_s_all_chromosomes": round(dt_rep, 3), "rows": rows, "covered": cov, "genome_bp": int(lens.sum())}))


if __name__ == "__main__":
    main()
