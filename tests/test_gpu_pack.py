"""RefSeq::Run_ConvertBinseq on the device (csrc/bsx_pack.hip, round 6): line-regular FASTA text is uploaded and packed by kernels, any other text goes
through the host packer — and both must give the oracle's words, anchors, chromosome tables and unmasked blocks (the oracle's packer is pinned to the real
reference's, tests/test_oracle_vs_reference.py).  The texts are those of tests/test_pack_cpu.py, written once with a uniform line width (device path)
and once with the seams the host packer's token rules exist for (CR LF, blank lines, ragged lines: host path)."""
import random

import numpy as np
import pytest

import bsmap_amd as B
from test_pack_cpu import _fasta

pytestmark = pytest.mark.gpu


def _sequences(seed, scale=1):
    """the records of test_pack_cpu._fasta as (name line, sequence) — N / X runs, IUPAC letters, short stretches, lower case"""
    rng = random.Random(seed)
    recs = []
    for c in range(rng.randint(2, 5)):
        n = rng.choice([9, 40, 3000, 20000, 60000]) * scale
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
        recs.append((">chr%d some description > with a mark" % (c + 1), "".join(s)[:n]))
    return recs


def _regular(recs, width, last_newline=True):
    out = []
    for i, (h, s) in enumerate(recs):
        lines = [s[k:k + width] for k in range(0, len(s), width)]
        out.append(h + "\n" + "\n".join(lines) + ("\n" if (last_newline or i + 1 < len(recs)) else ""))
    return "".join(out)


def _compare(gref, oref):
    f, c = gref.words()
    a, s, r = gref.info()
    assert np.array_equal(a, oref.anchor()) and np.array_equal(s, oref.chr_size()) and np.array_equal(r, oref.rc_offset())
    assert np.array_equal(f, oref.refcat()) and np.array_equal(c, oref.crefcat())
    assert np.array_equal(np.asarray(gref.blocks(), np.uint32).reshape(-1, 3), np.asarray(oref.blocks(), np.uint32).reshape(-1, 3))


@pytest.mark.parametrize("seed", list(range(8)))
@pytest.mark.parametrize("width", [60, 61, 7, 1000])
def test_line_regular_text_is_packed_on_the_device(seed, width, oracle):
    recs = _sequences(50 + seed, scale=2)
    while sum(len(s_) for _, s_ in recs) < (1 << 16):   # (the device packer takes texts of 64 KB and more)
        recs += [(h + "b", s_) for h, s_ in _sequences(500 + seed + len(recs), scale=2)]
    text = _regular(recs, width, last_newline=seed % 2 == 0)
    oref = oracle.OracleRef(oracle.make_params(), fasta_text=text, build_index=False)
    gref = B.RefSeq(B.make_params()).Run_ConvertBinseq(fasta_text=text)
    try:
        assert gref.packed_on_device
        _compare(gref, oref)
    finally:
        gref.close(); oref.free()


@pytest.mark.parametrize("seed", list(range(6)))
def test_any_other_text_takes_the_host_packer(seed, oracle, monkeypatch):
    """ragged lines, CR LF, blanks and tabs: the device finds the first violation and the host packer applies the reference's token rules; with BSX_HOST_PACK=1
    the regular text takes the host packer as well — same words"""
    text = _fasta(100 * seed + 3)
    while len(text) < (1 << 16):
        text += _fasta(100 * seed + 3 + len(text))
    oref = oracle.OracleRef(oracle.make_params(), fasta_text=text, build_index=False)
    gref = B.RefSeq(B.make_params()).Run_ConvertBinseq(fasta_text=text)
    try:
        assert not gref.packed_on_device
        _compare(gref, oref)
    finally:
        gref.close(); oref.free()
    reg = _regular(_sequences(70 + seed, scale=2), 60)
    monkeypatch.setenv("BSX_HOST_PACK", "1")
    oref = oracle.OracleRef(oracle.make_params(), fasta_text=reg, build_index=False)
    gref = B.RefSeq(B.make_params()).Run_ConvertBinseq(fasta_text=reg)
    try:
        assert not gref.packed_on_device
        _compare(gref, oref)
    finally:
        gref.close(); oref.free()


def test_rrbs_references_are_packed_on_the_host(oracle):
    text = _regular(_sequences(91, scale=2), 60)
    gref = B.RefSeq(B.make_params(D="C-CGG")).Run_ConvertBinseq(fasta_text=text)
    try:
        assert not gref.packed_on_device
    finally:
        gref.close()
