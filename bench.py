#!/usr/bin/env python3
"""bench.py — aligned reads/s of the HIP alignment hot path on BASELINE.json's headline configuration.

Workload (config.workload = "C3"): 2x150 bp (stored as 144 nt, the reference's READ_144 cap) paired-end WGBS reads,
-s 16 -v 6 -I 4 -m 28 -x 500, against an hg38-sized synthetic genome (24 sequences with hg38's chromosome lengths,
3.09 Gbp; generator in bsmap_amd/csrc/bsx_synth.hip, because hg38 itself is not on the GPU box).  A step is one
Do_Batch over --pairs-per-step read pairs that are already resident in HBM; value = reads (2 per pair) of all ranks /
wall time of the K timed steps (max over ranks).

  python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: spawns the N ranks itself, see launch_ranks)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU, reads sharded)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HG38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
        135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
        46709983, 50818468, 156040895, 57227415]
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes(c, n_reads):
    """8*N_lookup + sum_cand(4 + 8*W_c) + 80*N_orient + 16 per read"""
    n_lookup, n_cand, sum_w, n_orient = (int(x) for x in c[:4])
    return 8 * n_lookup + 4 * n_cand + 8 * sum_w + 80 * n_orient + 16 * n_reads


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start `python -m torch.distributed.run` with one rank
    per GPU as a CHILD process — before torch or libbsx is imported here, so this process never touches a GPU — relay its
    output (rank 0 prints the one JSON line) and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def selftest_launch():
    """what the ranks do under --selftest-launch (CPU test of the launcher path): gloo rendezvous, the same stats
    reduction as the real run, one JSON line from rank 0"""
    import torch.distributed as dist
    from bsmap_amd import sharding
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    mx, tot, table = sharding.gather_stats(0.5 * (rank + 1), [rank + 1, 10], dist)
    if rank == 0:
        print(json.dumps({"selftest": "launch", "n_gpus": world, "max_elapsed": mx, "sum": [float(x) for x in tot], "rows": int(table.shape[0])}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs-per-step", type=int, default=1 << 20)
    ap.add_argument("--genome", default="hg38", help="hg38 (3.09 Gbp synthetic, the bench config) or a fraction like 0.05 for quick checks")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="target CPU time of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--e2e-pairs", type=int, default=8 << 20, help="pairs of the end-to-end CLI run (FASTQ -> SAM in /dev/shm) reported beside the metric; 0 = skip")
    ap.add_argument("--in-flight", type=int, default=2, help="batches in flight per GPU (host threads, one device batch each): the latency-bound main "
                    "kernel of one batch overlaps the VALU-bound scan passes of the other; 1 = strictly one Do_Batch at a time")
    ap.add_argument("--waves-per-cu", type=int, default=0)
    ap.add_argument("--heavy-limits", default="", help="tuning: units per round,scan-task pool of the heavy pipeline (library default 32768,524288)")
    ap.add_argument("--heavy-threshold", type=int, default=0, help="tuning: candidate-list length that defers a unit to the heavy pipeline (0 = library default)")
    ap.add_argument("--mode", default="pe", choices=["pe", "se"], help="pe = C3 (default, the metric's config); se = C2 (1x100, -v 4)")
    ap.add_argument("--profile-serial", action="store_true", help="profiling mode: one batch in flight, one unit group (control and scan passes "
                    "strictly alternate), no CPU / end-to-end legs — no two kernels overlap, so per-kernel durations add up to at most the step time")
    ap.add_argument("--selftest-launch", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.selftest_launch:
        return selftest_launch()
    if args.profile_serial:
        args.in_flight, args.cpu_seconds, args.e2e_pairs = 1, 0.0, 0
        os.environ["BSX_HEAVY_GROUPS"] = "1"  # read by bsx_batch_create

    import torch  # first: libbsx.so then binds to the HIP runtime torch has already loaded
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or "MASTER_ADDR" in os.environ:  # launched by torch.distributed.run (also at N=1)
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.init()
    import numpy as np
    import bsmap_amd as B

    if args.waves_per_cu:
        B.lib().bsx_set_waves_per_cu(args.waves_per_cu)
    if args.heavy_threshold:
        B.lib().bsx_set_heavy_threshold(args.heavy_threshold)
    if args.heavy_limits:
        u_, t_ = (int(x) for x in args.heavy_limits.split(","))
        B.lib().bsx_set_heavy_limits(u_, t_)
    pe = args.mode == "pe"
    kw = dict(s=16, v=6, I=4, m=28, x=500, S=1, r=1, pairend=1) if pe else dict(s=16, v=4, I=4, S=1, r=1)
    read_len = 144 if pe else 100
    lens = HG38 if args.genome == "hg38" else [max(200_000, int(x * float(args.genome))) for x in HG38]
    t0 = time.time()
    real_fa = os.environ.get("BSX_HG38")  # a real genome FASTA if one is at hand (never on the driver's box); reads are
    if real_fa and os.path.exists(real_fa):  # still sampled on the device from the resident reference
        ref = B.RefSeq(B.make_params(**kw), device=local_rank).Run_ConvertBinseq(fasta_path=real_fa)
        lens = [int(x) for x in ref.info()[1]]
    else:
        real_fa = None
        ref = B.RefSeq(B.make_params(**kw), device=local_rank).synthetic(lens, seed=38)
    t_gen = time.time() - t0
    t0 = time.time()
    ref.CreateIndex()
    t_index = time.time() - t0
    B_ = args.pairs_per_step
    n_total = B_ * (args.steps + args.warmup)
    nfl = max(1, args.in_flight)
    batches = [(B.PairAlign if pe else B.SingleAlign)(ref, n_total) for _ in range(nfl)]
    batch = batches[0]
    # reads are sharded by rank: unit ids of rank r start at r * n_total (independent units, no data-path collective);
    # every device batch holds the same deterministic reads, step i runs on batch i % in_flight
    for bt in batches:
        bt.synth_reads(n_total, read_len, seed=3, first_index=rank * n_total)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    import threading

    def run_steps(lo, hi):
        def worker(j):
            for i in range(lo + j, hi, nfl):
                batches[j].run_range(i * B_, B_, sync=True)
                if i >= args.warmup:
                    kernel_ms.append(batches[j].kernel_ms())
                    scan_ms.append(batches[j].scan_ms())
        th = [threading.Thread(target=worker, args=(j,)) for j in range(1, nfl)]
        for t in th:
            t.start()
        worker(0)
        for t in th:
            t.join()

    kernel_ms, scan_ms = [], []
    run_steps(0, args.warmup)
    for bt in batches:
        bt.reset_counters()
    sync_all()
    t0 = time.perf_counter()
    run_steps(args.warmup, args.warmup + args.steps)
    sync_all()
    dt = time.perf_counter() - t0
    counters = sum(bt.counters().astype(np.float64) for bt in batches)
    reads_per_unit = 2 if pe else 1
    n_reads_rank = args.steps * B_ * reads_per_unit
    # stats reduction: the only collective of the path (RCCL all-gather of 9 doubles per rank)
    from bsmap_amd import sharding
    dt_max, tot_counters, allstats = sharding.gather_stats(dt, counters, dist, device="cuda")
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    value = n_reads_rank * world / dt_max
    alg_bytes_launch = algorithmic_bytes(counters, n_reads_rank) / args.steps
    # device time of one Do_Batch: HIP events on the batch's stream; with several batches in flight their kernels share the
    # GPU, so the step time that counts is the wall time per step of the timed region
    k_ms = float(np.mean(kernel_ms)) if nfl == 1 else dt_max / args.steps * 1e3
    achieved = alg_bytes_launch / (k_ms * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "aligned reads/sec (whole node), 2x150 bp hg38 WGBS -v 6 -s 16" if pe else "aligned reads/sec (whole node), 1x100 bp hg38 WGBS -v 4 -s 16",
        "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic" if not real_fa else "synthetic reads sampled from " + os.path.basename(real_fa),
        "config": {"workload": "C3: 2x150(->144) bp PE WGBS vs hg38-sized synthetic genome, -s 16 -v 6 -I 4 -m 28 -x 500" if pe
                   else "C2: 1x100 bp SE WGBS vs hg38-sized synthetic genome, -s 16 -v 4 -I 4",
                   "pairs_per_step" if pe else "reads_per_step": B_, "genome_bp": int(sum(lens)), "index_entries": int(ref.n_entries),
                   "parallelism": f"read-sharded x{world}", "batches_in_flight": nfl, "setup_s": {"genome": round(t_gen, 2), "index_build_gpu": round(t_index, 2)},
                   "aligned_fraction": float((2 * tot_counters[6] + tot_counters[5]) / max(1.0, n_reads_rank * world)) if pe
                   else float(tot_counters[5] / max(1.0, n_reads_rank * world))},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "kernel": "one Do_Batch = k_align + heavy pipeline (k_hctrl/k_hscan iterations)", "kernel_ms": k_ms,
                     "event_ms_per_do_batch": float(np.mean(kernel_ms)), "heavy_units_last_step": int(batch.heavy_units()), "algorithmic_bytes_per_launch": alg_bytes_launch,
                     "per_read": {"n_lookup": float(counters[0]) / n_reads_rank, "n_cand": float(counters[1]) / n_reads_rank,
                                  "ref_words64": float(counters[2]) / n_reads_rank},
                     "dominant_kernel": dominant_kernel(counters, scan_ms, args.steps)},
    }
    if world == 1 and args.cpu_seconds > 0:
        out["cpu_baseline"] = cpu_baseline(ref, batch, pe, kw, args.cpu_seconds, args.warmup * B_)
    for bt in batches:
        bt.close()
    ref.close()
    if world == 1 and pe and args.e2e_pairs > 0:
        out["end_to_end"] = end_to_end(args.e2e_pairs, args.genome)
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def dominant_kernel(counters, scan_ms, steps):
    """k_hscan, the kernel most of the time goes to: launches and HIP-event durations measured live (events on the stream it
    is launched on), algorithmic bytes of the candidates it evaluated (4 B index entry + 8 B per 64-bit reference word the
    reference's CountMismatch would touch).  Its reference gathers mostly hit the caches (clustered
    candidates), so the algorithmic rate may exceed what HBM delivers: the kernel is VALU-bound."""
    tot_ms = float(sum(t for t, n in scan_ms)); launches = int(sum(n for t, n in scan_ms))
    cand, words = float(counters[7]), float(counters[8])
    alg = 4.0 * cand + 8.0 * words
    if launches == 0 or tot_ms <= 0:
        return None
    return {"name": "k_hscan", "launches_per_step": launches / steps, "avg_launch_ms": tot_ms / launches, "ms_per_step": tot_ms / steps,
            "candidates_per_launch": cand / launches, "algorithmic_bytes_per_launch": alg / launches,
            "achieved_GBps": alg / (tot_ms * 1e-3) / 1e9, "frac_of_hbm_peak": alg / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "candidates_per_s": cand / (tot_ms * 1e-3), "bound": "valu (82 VALU instructions per 64 candidates, ~71 % of issue cycles)"}


def end_to_end(pairs, genome):
    """the same workload through the command-line driver — FASTA genome + two FASTQ files in, SAM out, all
    in /dev/shm — so that parsing, PCIe, formatting and writing are inside the time.  Reported beside `value`, never as it."""
    import types
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    try:
        import e2e_bench
        r = e2e_bench.run(types.SimpleNamespace(pairs=pairs, genome=1.0 if genome == "hg38" else float(genome), dir="/dev/shm/bsx_e2e_%d" % os.getpid(),
                                                threads=0, keep=False))
        t = r["timing"]
        return {"reads_per_s": r["reads_per_s_mapping_phase"], "unit": "reads/s, first batch parsed -> last SAM line written", "pairs": pairs,
                "mapping_s": t["mapping_s"], "load_reference_s": t["load_reference_s"], "index_build_s": t["index_build_s"], "whole_process_s": r["cli_wall_s"],
                "fastq_bytes": r["fastq_bytes"], "sam_bytes": r["sam_bytes"], "stage_busy_s": t["stage_busy_s"], "host_threads": t["workers"],
                "command": "bsmap -a r_1.fq -b r_2.fq -d genome.fa -o out.sam -s 16 -v 6 -m 28 -x 500 -S 1"}
    except Exception as e:  # the metric above does not depend on this leg
        return {"error": str(e)[:300]}


def cpu_baseline(ref, batch, pe, kw, target_s, first_unit):
    """the oracle (plain-C port of the reference algorithm, pthread batch model of main.cpp:49-73) timed on this box's
    host cores over a bounded sample of the SAME reads against the SAME reference + index (copied back from HBM)"""
    import numpy as np
    from oracle import oracle_ffi as O
    cores = os.cpu_count() or 1
    f, c = ref.words()
    a, s, r = ref.info()
    off, nf, ent = ref.index()
    oref = O.OracleRef.wrap(O.make_params(**kw), f, c, a, s, r, off, nf, ent)
    b1, o1 = batch.download_reads(0)
    L = int(o1[1] - o1[0])
    if pe:
        b2, o2 = batch.download_reads(1)

    def run(n):
        lo = first_unit
        oa = (o1[lo:lo + n + 1] - o1[lo]).copy()
        sa = b1[int(o1[lo]):int(o1[lo + n])].copy()
        t0 = time.perf_counter()
        if pe:
            sb = b2[int(o2[lo]):int(o2[lo + n])].copy()
            ob = (o2[lo:lo + n + 1] - o2[lo]).copy()
            t0 = time.perf_counter()
            O.pe_batch(oref, sa, oa, sb, ob, first_index=lo, threads=cores)
        else:
            O.se_batch(oref, sa, oa, first_index=lo, threads=cores)
        return time.perf_counter() - t0
    n0 = min(20000, len(o1) - 1 - first_unit)
    t_probe = run(n0)
    n = int(min(len(o1) - 1 - first_unit, max(n0, n0 * target_s / max(t_probe, 1e-3))))
    t = run(n)
    reads = n * (2 if pe else 1)
    return {"value": reads / t, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": f"{n} {'pairs' if pe else 'reads'} of the timed workload ({L} nt), oracle/bsx_oracle.c with {cores} pthreads, {t:.1f} s"}


if __name__ == "__main__":
    main()
