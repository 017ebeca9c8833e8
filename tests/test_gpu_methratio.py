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
        paths[c] = [str(d / fn) for fn in case["infiles"]]
    return fa, paths, d


@pytest.mark.parametrize("case,i", RUNS, ids=[f"{c}-{'_'.join(GOLD['cases'][c]['runs'][i]['options']) or 'default'}" for c, i in RUNS])
def test_methratio_matches_reference_script(case, i, files, capsys):
    from bsmap_amd import methratio
    fa, paths, d = files
    run = GOLD["cases"][case]["runs"][i]
    out = str(d / f"{case}_{i}.txt")
    methratio.main(["-q", "-o", out, "-d", fa] + list(run["options"]) + paths[case])
    assert open(out).read() == run["table"]
    stdout = capsys.readouterr().out
    if not run["crashed"]:
        assert stdout == run["stdout"]


def test_methratio_batches_do_not_matter(files):
    """duplicate removal is defined by input order: many small device batches give the same table as one"""
    from bsmap_amd import methratio
    fa, paths, d = files
    outs = []
    for b in (1 << 20, 97):
        out = str(d / f"b{b}.txt")
        s = methratio.run(fa, paths["pe"], out, rm_dup=True, meth0=True, combine_CpG=True, batch=b)
        outs.append((open(out).read(), s))
    assert outs[0] == outs[1] and outs[0][0].count("\n") > 1000
