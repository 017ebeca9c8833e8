"""bench.py's file path (BSX_HG38 = a genome FASTA) at size: the synthetic text of a ~300 Mb genome is written to a
FASTA; loading it through the text packer (RefSeq::Run_ConvertBinseq, dbseq.cpp:215-282) must give the very reference and index
the device-side generator gives, and `python bench.py` with BSX_HG38 pointing at the file must report the same workload
(index entries, work per read, aligned fraction) as the synthetic path on the same genome — its `data` field says which path ran."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import bsmap_amd as B

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FRACTION = "0.1"   # of hg38's chromosome lengths: 24 sequences, 309 Mb
KW = dict(s=16, v=6, I=4, m=28, x=500, S=1, r=1, pairend=1)


def _bench(env_extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--genome", FRACTION, "--steps", "2", "--warmup", "1", "--pairs-per-step", "65536", "--cpu-seconds", "0",
           "--e2e-pairs", "0", "--transfer-steps", "0", "--sensitivity", "0", "--other-configs", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, **env_extra))
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_on_a_fasta_equals_the_synthetic_path(tmp_path):
    import bench
    lens = [max(200_000, int(x * float(FRACTION))) for x in bench.HG38]
    syn = B.RefSeq(B.make_params(**KW)).synthetic(lens, seed=38).CreateIndex()
    fa = str(tmp_path / "genome.fa")
    with open(fa, "wb") as f:
        for c, nm in enumerate(syn.names()):
            f.write(f">{nm} synthetic\n".encode())
            t = syn.synth_bytes(c)
            for i in range(0, len(t), 1 << 24):   # (lines of 70 nt, as genome FASTA files have them)
                blk = t[i:i + (1 << 24)]
                full = len(blk) // 70 * 70
                f.write(np.concatenate([blk[:full].reshape(-1, 70), np.full((full // 70, 1), 10, np.uint8)], 1).tobytes())
                if full < len(blk):
                    f.write(blk[full:].tobytes() + b"\n")
    try:
        fil = B.RefSeq(B.make_params(**KW)).Run_ConvertBinseq(fasta_path=fa).CreateIndex()
        try:
            assert fil.names() == syn.names()
            for a, b in zip(syn.info(), fil.info()):
                assert np.array_equal(a, b)
            for a, b in zip(syn.words(), fil.words()):
                assert np.array_equal(a, b)
            assert syn.n_entries == fil.n_entries > 100_000_000
            for a, b in zip(syn.index(), fil.index()):
                assert np.array_equal(a, b)
        finally:
            fil.close()
    finally:
        syn.close()
    j_file = _bench({"BSX_HG38": fa})
    j_syn = _bench({"BSX_HG38": ""})
    assert "genome.fa" in j_file["data"] and j_syn["data"] == "synthetic"
    assert j_file["value"] > 0 and j_file["config"]["genome_bp"] == j_syn["config"]["genome_bp"] == sum(lens)
    assert j_file["config"]["index_entries"] == j_syn["config"]["index_entries"]
    assert j_file["config"]["aligned_fraction"] == j_syn["config"]["aligned_fraction"]
    assert j_file["roofline"]["per_read"] == j_syn["roofline"]["per_read"]
    d = os.path.join(ROOT, "gpurun_out", "validate")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "r04_bench_fasta.json"), "w") as f:
        json.dump({"fasta_path_line": {k: j_file[k] for k in ("metric", "value", "data", "ms_per_step", "config")},
                   "synthetic_path_line": {k: j_syn[k] for k in ("metric", "value", "data", "ms_per_step", "config")}}, f, indent=1)
