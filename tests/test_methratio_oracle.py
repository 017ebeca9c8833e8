"""The methylation-ratio oracle (oracle/methratio_oracle.py) against the output of the reference's own methratio.py on BSP
files written by the real bsmap binary (tests/golden/methratio.json.gz, made by tests/golden/make_golden_methratio.py)."""
import gzip
import json
import os

import pytest

import golden_util as G
from oracle import methratio_oracle as MO

GOLD = json.load(gzip.open(os.path.join(G.GOLDEN, "methratio.json.gz"), "rt"))
# (the BAM case is the paired SAM case in another container: the oracle reads text, the GPU tool decodes the BAM itself)
RUNS = [(c, i) for c in sorted(GOLD["cases"]) if not c.endswith("_bam") for i in range(len(GOLD["cases"][c]["runs"]))]


@pytest.mark.parametrize("case,i", RUNS, ids=[f"{c}-{'_'.join(GOLD['cases'][c]['runs'][i]['options']) or 'default'}" for c, i in RUNS])
def test_oracle_matches_reference_script(case, i):
    c = GOLD["cases"][case]
    run = c["runs"][i]
    table, summary = MO.run(GOLD["fasta"], [(f, c["files"][f]) for f in c["infiles"]], MO.options_from_argv(run["options"]))
    assert table == run["table"]
    if run["crashed"]:
        assert summary is None  # nothing covered: the reference divides by zero in its last print
    else:
        assert summary == run["stdout"]
