"""The RCCL branch of bench.py on hardware, as far as one GPU allows: `python -m torch.distributed.run --nproc-per-node 1 bench.py
--gpus 1 ...` initialises the process group with the nccl (= RCCL) backend before any libbsx call, runs the barrier-bracketed timed region, and
reduces time + counters with the one all-gather of the path (bsmap_amd/sharding.gather_stats — executed at world size 1, too).  The line must
equal the un-launched run's in everything but the clock."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--gpus", "1", "--genome", "0.02", "--pairs-per-step", "65536", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--e2e-pairs", "0", "--transfer-steps", "0",
        "--sensitivity", "0", "--other-configs", "0", "--work-counters", "1"]


def _line(out):
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


def test_bench_under_the_launcher_runs_the_rccl_branch_and_agrees_with_the_plain_run():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), BSX_TRACE_COLLECTIVE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py")] + ARGS
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    a = _line(r.stdout)
    # the process group was RCCL and the all-gather ran (bench.py says so on stderr when asked)
    assert "collective: backend nccl world 1 all_gather ok" in r.stderr, r.stderr[-2000:]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + ARGS, capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")})
    assert p.returncode == 0, p.stderr[-3000:]
    b = _line(p.stdout)
    assert a["n_gpus"] == b["n_gpus"] == 1 and a["config"]["parallelism"] == b["config"]["parallelism"] == "read-sharded x1"
    assert a["metric"] == b["metric"] and a["steps"] == b["steps"] == 2 and a["scaling"] == "weak"
    # same reads, same work: the counters behind the line are equal, the rates agree within the noise of so short a run
    for k in ("n_lookup", "n_cand", "ref_words64"):
        assert a["roofline"]["per_read"][k] == b["roofline"]["per_read"][k], k
    assert a["config"]["aligned_fraction"] == b["config"]["aligned_fraction"]
    assert a["roofline"]["algorithmic_bytes_per_launch"] == b["roofline"]["algorithmic_bytes_per_launch"]
    assert 0.5 < a["value"] / b["value"] < 2.0
