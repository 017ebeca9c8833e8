"""The reference packer of libbsx (csrc/bsx_host.cpp, chunk-parallel since round 6) against the oracle's packer — which is pinned to the real reference's
RefSeq::Run_ConvertBinseq (tests/test_oracle_vs_reference.py) — on FASTA texts built to hit the packer's seams: tokens split by blanks and tabs, N / X runs that
start, end and span chunk borders, IUPAC letters before the first ACGT of a stretch, stretches under 30 nt, lower case, CR LF line ends, '>' inside a line,
records shorter than a word.  The harness is built with chunk sizes of ~1 kb, so a 60 kb record crosses every border hundreds of times.  CPU only."""
import os
import random
import shutil
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc (host-only compile of csrc/bsx_host.cpp)")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("pk") / "pack_check")
    subprocess.run([HIPCC, "-O1", "-g", "-std=c++17", "-x", "hip", "--cuda-host-only", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "harness", "pack_check.cpp")],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    return exe


def _fasta(seed):
    rng = random.Random(seed)
    recs = []
    for c in range(rng.randint(2, 5)):
        n = rng.choice([9, 40, 3000, 20000, 60000])
        s = []
        while sum(len(x) for x in s) < n:
            kind = rng.random()
            if kind < 0.55:
                s.append("".join(rng.choice("ACGT") for _ in range(rng.randint(1, 1500))))
            elif kind < 0.75:
                s.append(rng.choice("NXnx") * rng.randint(1, 2500))
            elif kind < 0.85:
                s.append("".join(rng.choice("RYKMSWBDHV") for _ in range(rng.randint(1, 6))))
            elif kind < 0.93:
                s.append("".join(rng.choice("acgt") for _ in range(rng.randint(1, 45))))
            else:
                s.append("C" + rng.choice(["CGG", "cgg"]) * rng.randint(1, 3))
        seq = "".join(s)[:n]
        lines, i = [], 0
        while i < len(seq):
            w = rng.choice([60, 60, 60, 61, 7, 1, 200])
            lines.append(seq[i:i + w] + rng.choice(["\n", "\n", "\r\n", " \n", "\t\n", "\n\n"]))
            i += w
        recs.append(">chr%d some description > with a mark%s" % (c + 1, rng.choice(["\n", "\r\n"])) + "".join(lines))
    return "".join(recs)


def _read_dump(path, rrbs):
    b = open(path, "rb").read()
    nc, nw, nb = struct.unpack_from("<IQI", b, 0)
    o = 16
    def take(n):
        nonlocal o
        a = np.frombuffer(b, np.uint32, n, o); o += 4 * n
        return a
    d = {"anchor": take(nc + 1), "chr_size": take(nc), "rc_offset": take(nc), "refcat": take(nw), "crefcat": take(nw), "blocks": take(3 * nb).reshape(-1, 3)}
    if rrbs:
        d["sites"] = []
        for _ in range(nc):
            n = int(take(1)[0]); d["sites"].append(take(n))
    return d


@pytest.mark.parametrize("seed", list(range(12)))
@pytest.mark.parametrize("rrbs", [False, True], ids=["wgbs", "rrbs"])
def test_parallel_packer_equals_the_oracle(seed, rrbs, harness, oracle, tmp_path):
    text = _fasta(100 * seed + (7 if rrbs else 0))
    fa = tmp_path / "g.fa"
    fa.write_text(text, newline="")
    out = str(tmp_path / "dump.bin")
    subprocess.run([harness, str(fa), out] + (["C-CGG"] if rrbs else []), check=True, timeout=300)
    got = _read_dump(out, rrbs)
    kw = dict(D="C-CGG") if rrbs else {}
    oref = oracle.OracleRef(oracle.make_params(**kw), fasta_path=str(fa), build_index=False)
    try:
        for f in ("anchor", "chr_size", "rc_offset", "refcat", "crefcat"):
            assert np.array_equal(got[f], getattr(oref, f)()), f
        assert np.array_equal(got["blocks"], np.asarray(oref.blocks(), np.uint32).reshape(-1, 3))
        if rrbs:
            for c in range(len(got["sites"])):
                assert np.array_equal(got["sites"][c], oref.sites(c)), ("sites", c)
    finally:
        oref.free()
