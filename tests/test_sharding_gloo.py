"""The N>1 path on CPU: two gloo ranks shard a batch, run the oracle on their shard (stand-in for the per-GPU kernel,
which needs a device), and reduce the statistics exactly as bench.py does over RCCL."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from bsmap_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_partition_exactly():
    for n in (0, 1, 7, 1000, 1048577):
        for w in (1, 2, 3, 8):
            r = [sharding.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_gather():
    mx, tot, table = sharding.gather_stats(1.5, [3, 4, 5])
    assert mx == 1.5 and list(tot) == [3, 4, 5] and table.shape == (1, 4)
    assert sharding.whole_job_rate([10, 10], 2, 2.0) == 20.0


WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np, torch, torch.distributed as dist
    from bsmap_amd import sharding
    from oracle import oracle_ffi as O
    import golden_util as G
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    meta, arr, fasta = G.load("c2_se100_n1")
    oref = O.OracleRef(O.make_params(**meta["kw"]), fasta_path=fasta)
    seqs = [r["seq"][:144] for r in meta["reads"]]
    lo, hi = sharding.shard_range(len(seqs), rank, world)
    buf, off = O.pack_reads(seqs[lo:hi])
    res, cnt = O.se_batch(oref, buf, off, first_index=lo, threads=1)
    aligned = int(((res["n_best"] == 1) | ((res["n_best"] > 1) & (meta["kw"]["r"] == 1))).sum())
    mx, tot, table = sharding.gather_stats(0.25 * (rank + 1), cnt + [hi - lo, aligned], dist)
    if rank == 0:
        print("RESULT " + json.dumps(dict(max=mx, tot=[float(x) for x in tot], rows=table.shape[0], picks=res["loc"][:5].tolist())))
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_rank_gloo_matches_single_rank(tmp_path, oracle):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][0]
    got = json.loads(line[7:])
    # single-rank truth
    import golden_util as G
    meta, arr, fasta = G.load("c2_se100_n1")
    oref = oracle.OracleRef(oracle.make_params(**meta["kw"]), fasta_path=fasta)
    seqs = [r["seq"][:144] for r in meta["reads"]]
    buf, off = oracle.pack_reads(seqs)
    res, cnt = oracle.se_batch(oref, buf, off, threads=1)
    aligned = int(((res["n_best"] == 1) | (res["n_best"] > 1)).sum())
    assert got["rows"] == 2 and got["max"] == 0.5
    assert got["tot"] == [float(x) for x in cnt] + [float(len(seqs)), float(aligned)]
    oref.free()


def test_bench_launcher_spawns_ranks_itself():
    """`python bench.py --gpus 2` without torchrun: the launcher path of bench.py starts the two ranks as a child process
    (gloo here, RCCL on GPUs) and rank 0 prints exactly one JSON line"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launch"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    got = json.loads(lines[0])
    assert got == {"selftest": "launch", "n_gpus": 2, "max_elapsed": 1.0, "sum": [3.0, 20.0], "rows": 2}


def test_eight_rank_gloo_reduction(tmp_path):
    """the stats reduction at the node's full width (8 ranks, gloo): every rank's row arrives, the maximum time and the counter sums are exact, and the
    shards partition the batch — what bench.py --gpus 8 does over RCCL, without a device"""
    import json
    script = tmp_path / "worker8.py"
    script.write_text(textwrap.dedent("""
        import os, sys, json
        sys.path.insert(0, {root!r})
        import torch.distributed as dist
        from bsmap_amd import sharding
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        lo, hi = sharding.shard_range(1000003, rank, world)
        mx, tot, table = sharding.gather_stats(0.125 * (rank + 1), [hi - lo, lo, 1, rank], dist)
        if rank == 0:
            print("RESULT " + json.dumps(dict(max=mx, tot=[float(x) for x in tot], rows=int(table.shape[0]), times=[float(x) for x in table[:, 0]])))
        dist.barrier()
        dist.destroy_process_group()
    """).format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", str(script)], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    ranges = [sharding.shard_range(1000003, r, 8) for r in range(8)]
    assert got["rows"] == 8 and got["max"] == 1.0 and got["times"] == [0.125 * (r + 1) for r in range(8)]
    assert got["tot"] == [1000003.0, float(sum(lo for lo, _ in ranges)), 8.0, 28.0]


def test_rank_cpu_shares_are_disjoint_and_on_the_node():
    """bench.py pins each rank to its share of the quota on its GPU's NUMA node: ranks that share a node get consecutive, disjoint CPU ranges; a node with too
    few CPUs pins nothing"""
    sys.path.insert(0, ROOT)
    import bench
    node = set(range(64, 128))
    shares = [bench.rank_cpu_share(node, 4, k, 8) for k in range(4)]
    assert all(len(s) == 8 and s <= node for s in shares)
    assert len(set().union(*shares)) == 32 and min(shares[0]) == 64 and min(shares[3]) == 88
    assert bench.rank_cpu_share(set(range(6)), 4, 0, 2) is None and bench.rank_cpu_share(node, 4, 1, 0) is None
