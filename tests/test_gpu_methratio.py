"""bsmap_amd.methratio (Python host + HIP pile-up kernels through the C ABI) against the output of the reference's own
methratio.py (tests/golden/methratio.json.gz): table files byte-identical for every option set, summary line identical."""
import gzip
import json
import os

import pytest

import golden_util as G

pytestmark = pytest.mark.gpu
GOLD = json.load(gzip.open(os.path.join(G.GOLDEN, "methratio.json.gz"), "rt"))
RUNS = [(c, i) for c in sorted(GOLD["cases"]) for i in range(len(GOLD["cases"][c]["runs"]))]


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("meth")
    fa = str(d / "g.fa")
    open(fa, "w").write(GOLD["fasta"])
    paths = {}
    for c, case in GOLD["cases"].items():
        for fn, txt in case["files"].items():
            open(str(d / fn), "w").write(txt)
        for fn, b64 in case.get("files_b64", {}).items():
            import base64
            open(str(d / fn), "wb").write(base64.b64decode(b64))
        paths[c] = [str(d / fn) for fn in case["infiles"]]
    return fa, paths, d


@pytest.mark.parametrize("case,i", RUNS, ids=[f"{c}-{'_'.join(GOLD['cases'][c]['runs'][i]['options']) or 'default'}" for c, i in RUNS])
def test_methratio_matches_reference_script(case, i, files, capsys):
    from bsmap_amd import methratio
    fa, paths, d = files
    run = GOLD["cases"][case]["runs"][i]
    if "same_as" in run:  # the BAM file was converted from that case's SAM file by the vendored samtools: same expected output
        run = [r for r in GOLD["cases"][run["same_as"]]["runs"] if r["options"] == run["options"]][0]
    out = str(d / f"{case}_{i}.txt")
    methratio.main(["-q", "-o", out, "-d", fa] + list(run["options"]) + paths[case])
    assert open(out).read() == run["table"]
    stdout = capsys.readouterr().out
    if not run["crashed"]:
        assert stdout == run["stdout"]


def test_methratio_low_level_calls_agree_with_the_file_path(files):
    """bsx_meth_add with explicit arrays in many small batches (duplicate removal is defined by input order across calls)
    gives the same rows as the file-level call"""
    import ctypes as C
    import numpy as np
    from bsmap_amd import methratio, _check
    fa, paths, d = files
    out = str(d / "file.txt")
    methratio.run(fa, paths["pe"], out, rm_dup=True, meth0=True)
    L = methratio._bind()
    ref = methratio.load_reference(fa, [])
    names = list(ref)
    lens = np.array([len(ref[n]) for n in names], np.uint64)
    h = C.c_void_p()
    _check(L.bsx_meth_create(len(names), lens.ctypes.data, 1, 0, C.byref(h)))
    for i, n in enumerate(names):
        _check(L.bsx_meth_set_reference(h, i, ref[n].encode()))
    st = {"++": 0, "-+": 1, "+-": 2, "--": 3}
    rows = []
    for p_ in paths["pe"]:
        for line in open(p_):
            col = line.split("\t")
            if col[3][:2] in ("NM", "QC"):
                continue
            rows.append((names.index(col[4]), int(col[5]) - 1, st[col[6]], int(col[7]), col[1]))
    for b0 in range(0, len(rows), 53):
        part = rows[b0:b0 + 53]
        off = np.cumsum([0] + [len(r[4]) for r in part]).astype(np.uint64)
        seq = np.frombuffer(("".join(r[4] for r in part)).encode() + b"\0", np.uint8)
        arr = [np.array([r[0] for r in part], np.uint32), np.array([r[1] for r in part], np.int64), np.array([r[2] for r in part], np.uint8),
               np.array([r[3] for r in part], np.int32), np.full(len(part), -1, np.int64), seq, off]
        _check(L.bsx_meth_add(h, len(part), *[a.ctypes.data for a in arr], 2))
    got = []
    for i, n in enumerate(names):
        nr = C.c_uint32()
        _check(L.bsx_meth_report_chr(h, i, 1, 1, C.byref(nr), None, None))
        pos, dep, met = (np.zeros(nr.value, np.uint32) for _ in range(3))
        _check(L.bsx_meth_fetch_rows(h, pos.ctypes.data, dep.ctypes.data, met.ctypes.data))
        got += [(n, int(a) + 1, int(b), int(c)) for a, b, c in zip(pos, dep, met)]
    L.bsx_meth_destroy(h)
    exp = [(f[0], int(f[1]), int(f[5]), int(f[6])) for f in (l.split("\t") for l in open(out).read().split("\n")[1:] if l)]
    assert sorted(got) == sorted(exp) and len(got) > 1000


@pytest.mark.parametrize("window", ["1", "70000", "1000000"])
def test_methratio_bam_is_streamed_in_bounded_windows(window, files, capsys, monkeypatch):
    """the BAM path inflates a bounded window of BGZF blocks at a time (a whole-genome BAM does not fit host memory): with
    windows of one block, a few blocks, everything — header and records straddling the window edges — the table stays the
    reference's (duplicate removal across windows keeps the file order)"""
    from bsmap_amd import methratio
    fa, paths, d = files
    import bam_util
    case = [c for c in sorted(GOLD["cases"]) if any(p.endswith(".bam") for p in paths[c])][0]
    # the same file in BGZF blocks of 997 bytes: every window edge then cuts through the header or a record
    small = []
    for p_ in paths[case]:
        data, _ = bam_util.read_bgzf(p_)
        q = str(d / ("small_" + os.path.basename(p_)))
        with open(q, "wb") as f:
            for i in range(0, len(data), 997):
                f.write(bam_util._bgzf_block(data[i:i + 997]))
            f.write(bam_util._bgzf_block(b""))
        small.append(q)
    monkeypatch.setenv("BSX_BAM_WINDOW", window)
    for i, run in enumerate(GOLD["cases"][case]["runs"][:6]):
        exp = run
        if "same_as" in run:
            exp = [r for r in GOLD["cases"][run["same_as"]]["runs"] if r["options"] == run["options"]][0]
        out = str(d / f"win_{window}_{i}.txt")
        methratio.main(["-q", "-o", out, "-d", fa] + list(run["options"]) + small)
        assert open(out).read() == exp["table"]
        capsys.readouterr()
