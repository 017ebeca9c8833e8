"""The command-line driver's memory-mapped FASTA/FASTQ reader (bsmap_amd/csrc/bsx_reads.h) against an iostream restatement
of the reference's reader (operator>> / getline(ch, 1000) token rules, reads.cpp:83-117) on well-formed and hostile files."""
import os
import random
import subprocess

import pytest

from conftest import HOST_SAN_FLAGS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("rh") / "reader_check")
    subprocess.run(["g++"] + HOST_SAN_FLAGS + ["-o", exe, os.path.join(ROOT, "tests", "harness", "reader_check.cpp"), "-lz"], check=True)   # (ASan + UBSan: conftest.py)
    return exe


def _run(exe, path, batch=7, start=1, end=4294967295, maxlen=144):
    out = subprocess.run([exe, path, str(batch), str(start), str(end), str(maxlen)], capture_output=True, timeout=60).stdout
    fast, stream = out.split(b"== stream\n")
    return fast.replace(b"== fast\n", b""), stream


def _fastq(rng, n, eol="\n", tail=True, odd=False):
    recs = []
    for i in range(n):
        L = rng.randint(1, 200)
        seq = "".join(rng.choice("ACGTN") for _ in range(L))
        qual = "".join(chr(rng.randint(33, 73)) for _ in range(L))
        name = f"read{i}" + (f" desc {i}\tmore" if odd and i % 3 == 0 else "")
        plus = "+" + (name if odd and i % 4 == 0 else "")
        gap = eol * rng.randint(0, 2) if odd else ""
        lead = " " * rng.randint(0, 2) if odd else ""
        recs.append(f"{lead}@{name}{eol}{seq}{eol}{plus}{eol}{qual}{eol}{gap}")
    txt = "".join(recs)
    return txt if tail else txt.rstrip("\r\n")


CASES = {
    "plain": lambda r: _fastq(r, 50),
    "crlf": lambda r: _fastq(r, 40, eol="\r\n"),
    "no_final_newline": lambda r: _fastq(r, 23, tail=False),
    "odd_layout": lambda r: _fastq(r, 60, odd=True),
    "odd_crlf_no_tail": lambda r: _fastq(r, 31, eol="\r\n", tail=False, odd=True),
    "truncated_record": lambda r: _fastq(r, 10) + "@last\nACGT\n",
    "name_only": lambda r: _fastq(r, 5) + "@lonely",
    "long_header": lambda r: _fastq(r, 6) + "@x " + "h" * 1500 + "\nACGT\n+\nIIII\n" + _fastq(r, 3),
    "header_999": lambda r: "@a " + "h" * 996 + "\nACGT\n+\nIIII\n" + _fastq(r, 3),
    "header_1000": lambda r: "@a " + "h" * 997 + "\nACGT\n+\nIIII\n" + _fastq(r, 3),
    "control_bytes": lambda r: "@c1\nAC\x01GT\x7f\n+\nII\x02III\n@c2\nAC\xc3\xa9GT\n+\nIIIIII\n",
    "fasta": lambda r: "".join(f">s{i} d\n{''.join(r.choice('ACGT') for _ in range(r.randint(1, 180)))}\n" for i in range(30)),
    "fasta_no_tail": lambda r: ">a\nACGT\n>b\nGGCC",
    "single": lambda r: "@only\nACGTACGT\n+\nIIIIIIII\n",
    "marker_space_name": lambda r: "@ spaced\nACGT\n+\nIIII\n@\ttabbed\nGG\n+\nII\n",
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_reader_token_rules(harness, case, tmp_path):
    rng = random.Random(hash(case) & 0xffff)
    path = str(tmp_path / "r.txt")
    with open(path, "w", newline="", encoding="latin-1") as f:
        f.write(CASES[case](rng))
    for batch, start, end, maxlen in ((7, 1, 4294967295, 144), (1000, 1, 4294967295, 50), (3, 4, 17, 144), (5, 2, 2, 100)):
        fast, stream = _run(harness, path, batch, start, end, maxlen)
        assert fast == stream, (case, batch, start, end)
        assert b"-- batch" in fast


def test_reader_large_random(harness, tmp_path):
    rng = random.Random(99)
    path = str(tmp_path / "big.fq")
    with open(path, "w", newline="") as f:
        f.write(_fastq(rng, 5000, odd=True))
    fast, stream = _run(harness, path, 512)
    assert fast == stream and fast.count(b"\n") > 5000


def test_reader_bam_records(harness, tmp_path):
    """BAM input (BGZF blocks + alignment records) read natively: names, bases (4-bit codes), qualities (+33), truncation to
    the read-length cap, record boundaries that straddle BGZF blocks; paired mode takes alternating records"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bam_util
    rng = random.Random(5)
    reads = []
    for i in range(300):
        L = rng.randint(1, 180)
        reads.append((f"r{i}", "".join(rng.choice("ACGTN") for _ in range(L)), "".join(chr(rng.randint(33, 73)) for _ in range(L))))
    path = str(tmp_path / "r.bam")
    bam_util.write_bam(path, [bam_util.record(n, s, q) for n, s, q in reads], block=997)
    fast, stream = _run(harness, path, batch=64, maxlen=144)
    lines = [l.split(b"\t") for l in fast.split(b"\n") if l and not l.startswith(b"--")]
    assert len(lines) == len(reads)
    for (i, (n, s, q)), l in zip(enumerate(reads), lines):
        assert (int(l[0]), l[1].decode(), l[2].decode(), l[3].decode()) == (i, n, s[:144], q[:144])
    # -B 5 -E 20 on BAM: the index starts at 4, no record is skipped (the reference's behaviour)
    fast, _ = _run(harness, path, batch=64, start=5, end=20, maxlen=144)
    lines = [l.split(b"\t") for l in fast.split(b"\n") if l and not l.startswith(b"--")]
    assert [int(l[0]) for l in lines] == list(range(4, 20)) and lines[0][1] == b"r0"
