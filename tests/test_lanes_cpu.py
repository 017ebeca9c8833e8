"""bsx_lanes.h (how `bsmap --lanes` cuts the read files: records of 4 / 2 lines, byte offsets from a parallel newline count,
every cut checked to sit on a record start) against a line-by-line restatement in Python.  The rule is the reference's own
`-B` skip: (read_start - 1) * 4 lines by getline (reads.cpp:50-76)."""
import json
import os
import random
import subprocess

import pytest

from conftest import HOST_SAN_FLAGS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("lh") / "lanes_check")
    subprocess.run(["g++"] + HOST_SAN_FLAGS + ["-o", exe, os.path.join(ROOT, "tests", "harness", "lanes_check.cpp")], check=True)   # (ASan + UBSan: conftest.py)
    return exe


def _plan(exe, a, b, lanes, start=1, end=4294967295):
    out = subprocess.run([exe, a, b or "-", str(lanes), str(start), str(end)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout)


def _fastq(path, n, rng, tail=True, qual_at=False):
    offs = []
    with open(path, "w") as f:
        pos = 0
        for i in range(n):
            L = rng.randint(1, 150)
            q = "".join(chr(rng.randint(35, 73)) for _ in range(L))
            if qual_at and i % 3 == 0:
                q = "@" + q[1:]   # a quality line may begin with '@'
            rec = f"@r{i} x{rng.randint(0, 10 ** rng.randint(0, 6))}\n{''.join(rng.choice('ACGTN') for _ in range(L))}\n+\n{q}\n"
            if i == n - 1 and not tail:
                rec = rec[:-1]
            offs.append(pos)
            f.write(rec)
            pos += len(rec)
    return offs


def test_cuts_fall_on_the_right_records(harness, tmp_path):
    rng = random.Random(5)
    a, b = str(tmp_path / "a.fq"), str(tmp_path / "b.fq")
    n = 12345
    oa, ob = _fastq(a, n, rng, qual_at=True), _fastq(b, n, rng, tail=False)
    for lanes in (1, 2, 3, 7, 64):
        p = _plan(harness, a, b, lanes)
        assert p["why_not"] == "" and p["total"] == n and p["n_a"] == p["n_b"] == n and len(p["lanes"]) == lanes
        nxt = 0
        for first, count, off_a, off_b in p["lanes"]:
            assert first == nxt and off_a == oa[first] and off_b == ob[first]
            nxt = first + count
        assert nxt == n
        sizes = [c for _, c, _, _ in p["lanes"]]
        assert max(sizes) - min(sizes) <= 1
    # -B / -E select a sub-range, which is what gets cut
    p = _plan(harness, a, b, 4, start=1001, end=9000)
    assert p["total"] == 8000 and p["lanes"][0][0] == 1000 and p["lanes"][-1][0] + p["lanes"][-1][1] == 9000
    assert [l[2] for l in p["lanes"]] == [oa[l[0]] for l in p["lanes"]]
    # more lanes than reads
    p = _plan(harness, a, None, 50, start=12340)
    assert len(p["lanes"]) == 6 and p["total"] == 6


def test_unequal_mates_follow_the_references_batches_of_50000(harness, tmp_path):
    rng = random.Random(6)
    a, b = str(tmp_path / "a.fq"), str(tmp_path / "b.fq")
    _fastq(a, 120_000, rng)
    _fastq(b, 119_990, rng)
    p = _plan(harness, a, b, 3)
    assert p["mates_differ"] and p["total"] == 100_000 and sum(l[1] for l in p["lanes"]) == 100_000
    p = _plan(harness, a, b, 3, end=30_000)   # the shorter file is never reached: nothing differs
    assert not p["mates_differ"] and p["total"] == 30_000
    _fastq(b, 40_000, rng)
    p = _plan(harness, a, b, 3)
    assert p["lanes"] == [] and "50000" in p["why_not"]


def test_files_that_cannot_be_cut(harness, tmp_path):
    rng = random.Random(7)
    a = str(tmp_path / "a.fq")
    _fastq(a, 100, rng)
    txt = open(a).read().split("\n")
    odd = str(tmp_path / "odd.fq")
    open(odd, "w").write("\n".join(txt[:40] + [""] + txt[40:]))   # a blank line inside: records are no longer 4 lines each
    assert _plan(harness, odd, None, 4)["lanes"] == [] and "record start" in _plan(harness, odd, None, 4)["why_not"]
    assert len(_plan(harness, odd, None, 1)["lanes"]) == 1   # one lane = nothing to check: the reader's token rules apply as ever
    lead = str(tmp_path / "lead.fq")
    open(lead, "w").write("\n" + open(a).read())
    assert _plan(harness, lead, None, 2)["lanes"] == []
    fa = str(tmp_path / "r.fa")
    with open(fa, "w") as f:
        for i in range(999):
            f.write(f">s{i}\n{'ACGT' * rng.randint(4, 30)}\n")
    p = _plan(harness, fa, None, 4)
    assert p["total"] == 999 and len(p["lanes"]) == 4
    assert _plan(harness, fa, a, 2)["lanes"] == []   # FASTA with FASTQ mates
    empty = str(tmp_path / "e.fq")
    open(empty, "w").close()
    assert _plan(harness, empty, None, 2)["lanes"] == []


def test_exact_mode_lane_states_compose_word_by_word(harness):
    """BSX_P1_EXACT with lanes: lane l starts from the effect of the nearest earlier lane that wrote each word of the reference's never-reset planner state,
    zero (a fresh object) if none did; a lane that delivered no effect changes nothing.  Checked against a sequential replay in Python."""
    import random
    rng = random.Random(7)
    for W, L in ((1, 1), (3, 4), (5, 8)):
        for _ in range(20):
            lanes = []
            for l in range(L):
                lanes.append(["-"] * W if rng.random() < 0.15 else [("x" if rng.random() < 0.5 else str(rng.randrange(0, 2**32 - 1))) for _ in range(W)])
            out = subprocess.run([harness, "compose", str(W), str(L)] + [t for row in lanes for t in row], capture_output=True, text=True, check=True)
            got = json.loads(out.stdout)
            st, want = [0] * W, []
            for row in lanes:
                want.append(list(st))
                if row[0] != "-":
                    st = [s if t == "x" else int(t) for s, t in zip(st, row)]
            assert got == want
