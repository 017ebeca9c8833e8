#!/usr/bin/env python3
"""bench.py — aligned reads/s of the HIP alignment hot path on BASELINE.json's configurations.

Default workload (config.workload = "C3", the one the metric is quoted on): 2x150 bp (stored as 144 nt, the reference's
READ_144 cap) paired-end WGBS reads, -s 16 -v 6 -I 4 -m 28 -x 500, against an hg38-sized synthetic genome (24 sequences
with hg38's chromosome lengths, 3.09 Gbp; generator in bsmap_amd/csrc/bsx_synth.hip, because hg38 itself is not on the
GPU box).  A step is one Do_Batch over --pairs-per-step units that are already resident in HBM; value = reads of all
ranks / wall time of the K timed steps (max over ranks).  --mode se | rrbs | trim run C2 / C4 / C5 the same way.

  python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: spawns the N ranks itself, see launch_ranks)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU, reads sharded)
  python bench.py --profile-serial                      (profiling: no two kernels overlap; see profiles/README.md)
"""
import argparse
import hashlib
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
ADAPTER = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"

# BASELINE.json configs[1..4] (configs[0] is the CPU plumbing case, a parity test): options, read length, sampler kind
MODES = {
    "pe": dict(tag="C3", kw=dict(s=16, v=6, I=4, m=28, x=500, S=1, r=1, pairend=1), pe=True, L=144, kind=0,
               metric="aligned reads/sec (whole node), 2x150 bp hg38 WGBS -v 6 -s 16",
               workload="C3: 2x150(->144) bp PE WGBS vs hg38-sized synthetic genome, -s 16 -v 6 -I 4 -m 28 -x 500"),
    "se": dict(tag="C2", kw=dict(s=16, v=4, I=4, S=1, r=1), pe=False, L=100, kind=0,
               metric="aligned reads/sec (whole node), 1x100 bp hg38 WGBS -v 4 -s 16",
               workload="C2: 1x100 bp SE WGBS vs hg38-sized synthetic genome, -s 16 -v 4 -I 4"),
    "rrbs": dict(tag="C4", kw=dict(D="C-CGG", S=1, r=1), pe=False, L=75, kind=2,
                 metric="aligned reads/sec (whole node), 1x75 bp hg38 RRBS -D C-CGG (seed 12, interval 1)",
                 workload="C4: 1x75 bp RRBS reads at C-CGG sites vs hg38-sized synthetic genome, -D C-CGG (-s 12 -I 1 forced), -v 2"),
    "trim": dict(tag="C5", kw=dict(s=16, v=6, I=4, m=28, x=500, S=1, r=1, pairend=1, q=20, A=[ADAPTER]), pe=True, L=144, kind=1,
                 metric="aligned reads/sec (whole node), 2x150 bp hg38 WGBS with 3' adapters, -A <adapter> -q 20",
                 workload="C5: 2x150(->144) bp PE WGBS with low-quality 3' tails and adapter read-through, -s 16 -v 6 -m 28 -x 500 -q 20 -A " + ADAPTER),
}


RING_UNITS = 1 << 23    # resident reads of a device batch: a ring of distinct steps, 2^23 units in all (8 steps of 2^20), at least 4 steps (see main)


def ring_steps(steps, warmup, B_):
    return max(1, min(steps + warmup, max(4, RING_UNITS // B_)))


HBM_BYTES = 288e9       # MI355X


RRBS_POOLS = "160000,1900000"   # (round 6: 110000,1400000 until the control passes got cheaper — fewer, larger rounds: C4 415-420 -> 398 ms per step at 150000,1800000, gpurun_out/r06s; three batches of these pools plan 0.88 of the device)
# units per step, batches in flight and starting pools by mode (measured: profiles/r05d_*).  A larger device batch gives the scan kernel larger
# groups (more reads over one window and offset per pass): C3 24.3 M reads/s at 2^20 pairs per step with three in flight, 27.0 M at 2^22 with two.
# (profiles/r05e: C2 16.8 M at 2^20 x 3, 18.9 M at 2^22 x 2; C5 9.2 M at 2^20 x 3, 10.5 M at 2^22 x 3; C4 8.6 M at 2^20 x 3, 9.4 M at 2^22 x 3)
# (gpurun_out/r05o, final scan kernel: C3 at 2^22 pairs 299-301 ms per step with two batches in flight, 285.9 with three; 2^22 x 3 plans 0.87 of the device)
# With the context prefilter (23.6 GB of context words beside the index) the main kernel is 1.7 x faster and a third batch in flight no longer pays for C3
# (gpurun_out/r05s: 242.0-244.2 ms per step with two, 241.9 with three) — and 2^22 pairs x 3 would plan 0.95 of the device.
# (gpurun_out/r05t: C5 10.7 M reads/s at 2^22 x 2, 10.9 M at 3 x 2^20 x 3, 10.1 M at 2^21 x 3; C2 20.8 M at 2^22 x 2, 22.1 M at 2^22 x 3)
TRIM_POOLS = "31457,2500000"   # (round 6: C5's first passes ask for 2.4 M scan tasks — short trimmed reads publish whole windows that survive — where the library's starting size for 3 x 2^20 pairs is 1.57 M: refused requests cost their units a pass; 490 -> 468-470 ms per step, gpurun_out/r06u)
MODE_DEFAULTS = {"pe": (1 << 22, 2, None), "se": (1 << 22, 3, None), "trim": (3 << 20, 3, TRIM_POOLS), "rrbs": (1 << 22, 3, RRBS_POOLS)}


def mode_defaults(mode):
    """(units per step, batches in flight, starting pools (units per round, tasks) or None = the library's for that step size)"""
    b, f, lim = MODE_DEFAULTS[mode]
    return b, f, (tuple(int(x) for x in lim.split(",")) if lim else None)


def transfer_batches(nfl, rrbs, ref_bytes, batch_bytes):
    """device batches of the PCIe-inclusive leg: two more than the resident run keeps in flight (a batch that is moving data does not compute:
    3 / 4 batches 0.93 / 0.96 of the resident rate at 2^20 pairs per step), at most 4, and no more than fit 0.85 of the device (RRBS: three of 62 GB)"""
    return int(max(1, min(nfl + 2, 4, (0.85 * HBM_BYTES - ref_bytes) // batch_bytes)))


def memory_plan(B, params, pe, B_, steps, warmup, nfl, n_entries, transfers, rrbs, genome_bp=3.1e9, n_cu=256, blocks_per_cu=5, limits=None):
    """device bytes of each phase of a run, from host arithmetic alone (bsx_batch_plan_bytes; tests/test_bench_memory_cpu.py holds every
    mode's plan for the driver's `--steps 20 --warmup 5` under 0.9 of the device).  Pools count at their starting size: a batch halves them
    when less is free, so the plan is an upper bound."""
    ring = ring_steps(steps, warmup, B_)
    if limits is None:   # (what main() sets: pools for a step, not for the ring a batch holds; left set — bsx_set_heavy_limits(0, 0) restores the defaults)
        limits = B.default_heavy_limits(params, B_, pe)
    B.lib().bsx_set_heavy_limits(*limits)
    big = B.plan_bytes(params, B_ * ring, pe, n_entries, n_cu, blocks_per_cu)
    small = B.plan_bytes(params, B_, pe, n_entries, n_cu, blocks_per_cu)
    tot = lambda d: d["per_unit"] + d["scratch"] + d["pools"]
    # reference: packed copy + plane copy (4 bits per nt in all), bucket offsets and forward counts, entries (RRBS: {tag, loc} pairs + group offsets)
    K = int(params.total_kmers)
    ref_b = genome_bp / 16 * 4 * 2 * 2 + 8.0 * K + n_entries * (8 if rrbs else 4) + (K * 32 * 4 if rrbs else 0)
    if not rrbs and int(params.index_interval) <= 4 and os.environ.get("BSX_CTX") != "0":
        ref_b += 16.0 * n_entries   # the entries' context words (the main kernel's prefilter)
    build_b = 0 if rrbs else n_entries * 8 * 2 + n_entries * 4   # index build: key/value double buffers of the radix sort (transient)
    nt = transfer_batches(nfl, rrbs, ref_b, tot(small)) if transfers else 0
    # (the serial / counted replays and the CPU baseline's downloads use the one timed batch that is kept; it is closed before the PCIe-inclusive leg creates its own)
    phases = {"index_build": ref_b + build_b, "timed": ref_b + nfl * tot(big), "side_legs": ref_b + max(tot(big), nt * tot(small))}
    return {"phases_GB": {k: round(v / 1e9, 1) for k, v in phases.items()}, "peak_GB": round(max(phases.values()) / 1e9, 1),
            "peak_frac_of_device": round(max(phases.values()) / HBM_BYTES, 3), "per_batch_GB": {k: round(v / 1e9, 2) for k, v in big.items()},
            "transfer_leg_batches": nt, "transfer_leg_batch_GB": round(tot(small) / 1e9, 1), "reference_GB": round(ref_b / 1e9, 1)}


def algorithmic_bytes(c, n_reads):
    """8*N_lookup + sum_cand(4 + 8*W_c) + 80*N_orient + 16 per read"""
    n_lookup, n_cand, sum_w, n_orient = (int(x) for x in c[:4])
    return 8 * n_lookup + 4 * n_cand + 8 * sum_w + 80 * n_orient + 16 * n_reads


def lib_sha16():
    import bsmap_amd as B
    h = hashlib.sha256()
    with open(B.LIB_PATH, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:16]


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
    ap.add_argument("--steps", type=int, default=12)   # a multiple of 2, 3 and 4: every batch in flight runs the same number of steps
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs-per-step", type=int, default=0, help="units (pairs, or single reads) per step; 0 = the mode's default (MODE_DEFAULTS: 2^22 for C3)")
    ap.add_argument("--genome", default="hg38", help="hg38 (3.09 Gbp synthetic, the bench config) or a fraction like 0.05 for quick checks")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of each cpu_baseline sample (0 = skip)")
    ap.add_argument("--e2e-pairs", type=int, default=16 << 20, help="pairs of the end-to-end CLI run (FASTQ -> SAM in /dev/shm) reported beside the metric; 0 = skip")
    ap.add_argument("--in-flight", type=int, default=0, help="batches in flight per GPU (host threads, one device batch each): the main kernel and the "
                    "latency-bound tail of one batch overlap the scan passes of the other; 1 = strictly one Do_Batch at a time; 0 = the mode's default (MODE_DEFAULTS)")
    ap.add_argument("--transfer-steps", type=int, default=-1, help="steps of the PCIe-inclusive leg (upload -> Do_Batch -> results per step); -1 = as many as --steps (the same window length as the metric's), 0 = skip")
    ap.add_argument("--waves-per-cu", type=int, default=0)
    ap.add_argument("--heavy-limits", default="", help="tuning: units per round,scan-task pool of the heavy pipeline")
    ap.add_argument("--heavy-threshold", type=int, default=0, help="tuning: candidate-list length that defers a unit to the heavy pipeline (0 = library default)")
    ap.add_argument("--mode", default="pe", choices=sorted(MODES), help="pe = C3 (default, the metric's config); se = C2; rrbs = C4; trim = C5")
    ap.add_argument("--profile-serial", action="store_true", help="profiling mode: one batch in flight, one unit group (control and scan passes "
                    "strictly alternate), no CPU / end-to-end / transfer legs — no two kernels overlap, so per-kernel durations add up to at most the step time")
    ap.add_argument("--exact", action="store_true", help="run with bsx_batch_set_leak_exact (the single-threaded reference's planner state for every read): "
                    "the line then includes the pre-pass that finds that state (k_leak_meta / k_leak_resolve)")
    ap.add_argument("--sensitivity", type=int, default=1, help="1: also run the workload on two variants of the synthetic genome (microsatellite share halved; no repeat "
                    "elements) — the headline depends on the generator's repeat content; 0 = skip")
    ap.add_argument("--other-configs", type=int, default=1, help="1 (with the default mode, one GPU): after the metric's own config, run BASELINE's other single-GPU configs (C2, C4, C5) "
                    "for a few steps each in child processes and report them under other_configs — parity-test cases, not the metric")
    ap.add_argument("--work-counters", type=int, default=0, help="0 (default): the timed region runs as the command line does, with the work counters off (bsx_batch_set_work_counters: the scan "
                    "kernels skip the early-out classification that only the counters need; records identical) and the counters of the SURVEY 8(d) formula come from a counted "
                    "pass over the same resident steps afterwards; 1: the timed region itself is counted")
    ap.add_argument("--selftest-launch", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.selftest_launch:
        return selftest_launch()
    d_units, d_nfl, d_lim = mode_defaults(args.mode)
    args.units_given = args.pairs_per_step > 0
    if args.pairs_per_step <= 0:
        args.pairs_per_step = d_units
    if args.in_flight <= 0:
        args.in_flight = d_nfl
    if d_lim and not args.heavy_limits and not args.profile_serial and args.in_flight >= 3:   # (C4: three batches in flight only fit with two-thirds pools)
        args.heavy_limits = "%d,%d" % d_lim
    if args.profile_serial:
        args.in_flight, args.cpu_seconds, args.e2e_pairs, args.transfer_steps = 1, 0.0, 0, 0
        os.environ["BSX_HEAVY_GROUPS"] = "1"  # read by bsx_batch_create (also the default)

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

    rank_cpus = pin_rank_to_gpu_node(B, local_rank, world) if world > 1 else None
    if args.waves_per_cu:
        B.lib().bsx_set_waves_per_cu(args.waves_per_cu)
    if args.heavy_threshold:
        B.lib().bsx_set_heavy_threshold(args.heavy_threshold)
    M = MODES[args.mode]
    pe, kw, read_len = M["pe"], M["kw"], M["L"]
    # a device batch here HOLDS a ring of steps and runs one step at a time: its pools are sized for a step, not for what it holds
    u_, t_ = (int(x) for x in args.heavy_limits.split(",")) if args.heavy_limits else B.default_heavy_limits(B.make_params(**kw), args.pairs_per_step, pe)
    B.lib().bsx_set_heavy_limits(u_, t_)
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
    # Resident reads: a ring of distinct steps, 2^23 units or four steps (8 steps of 2^20 pairs, 4 of 2^22: a step's reads are 300 MB - 1.2 GB against
    # 4 MB of L2 and 256 MB of Infinity Cache — a step that comes round again finds nothing of itself anywhere).  Device memory then does not grow
    # with --steps (round 4: 25 steps x 3 batches of resident reads left no room for the side legs under the driver's own command).
    ring = ring_steps(args.steps, args.warmup, B_)
    n_total = B_ * ring
    nfl = max(1, args.in_flight)
    Align = B.PairAlign if pe else B.SingleAlign
    plan = memory_plan(B, B.make_params(**kw), pe, B_, args.steps, args.warmup, nfl, int(ref.n_entries), args.transfer_steps != 0 and world == 1, bool(kw.get("D")), float(sum(lens)), limits=(u_, t_))
    t0 = time.time()
    batches = [Align(ref, n_total) for _ in range(nfl)]
    for bt in batches:
        bt.set_work_counters(bool(args.work_counters))
        if args.exact:
            bt.set_leak_exact()
    t_batches = time.time() - t0
    batch = batches[0]
    # reads are sharded by rank: unit ids of rank r start at r * n_total (independent units, no data-path collective);
    # every device batch holds the same deterministic reads, step i runs on batch i % in_flight
    for bt in batches:
        bt.synth_reads(n_total, read_len, seed=3, first_index=rank * n_total, kind=M["kind"])

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    import threading

    def run_steps(lo, hi):
        def worker(j):
            for i in range(lo + j, hi, nfl):
                batches[j].run_range((i % ring) * B_, B_, sync=True)
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
    serial_err = None
    run_steps(0, args.warmup)
    for bt in batches:
        bt.reset_counters()
    sync_all()
    t0 = time.perf_counter()
    run_steps(args.warmup, args.warmup + args.steps)
    sync_all()
    dt = time.perf_counter() - t0
    counters = sum(bt.counters().astype(np.float64) for bt in batches)   # (with the work counters off: completed by the counted pass below)
    heavy_last = [int(batch.heavy_units()), int(batch.redo_units())]
    # per-kernel evidence comes from a serial replay (one batch, control and scan passes strictly alternating): with batches in
    # flight the launches of different batches overlap and their durations say nothing about one kernel
    pools = [bt.pool_sizes() for bt in batches]
    serial = None
    for bt in batches[1:]:   # the side legs below need the memory; batches[0] keeps the resident reads they sample
        bt.close()
    batches = batches[:1]
    def serial_steps(b_, slots):
        ms, sc, st = [], [], []
        b_.set_stage_timing(True)   # (three more events per control pass: only here)
        try:
            t1 = time.perf_counter()
            for k in slots:
                b_.run_range(k * B_, B_, sync=True)
                ms.append(b_.kernel_ms()); sc.append(b_.scan_ms()); st.append(b_.stage_ms())
            wall = (time.perf_counter() - t1) / len(slots) * 1e3
        finally:
            b_.set_stage_timing(False)
        stage = {k: float(np.mean([x[k] for x in st])) for k in ("k_align", "k_hctrl", "order", "scan")}
        stage["control_passes"] = float(np.mean([x["control_passes"] for x in st]))
        return {"ms_per_step": wall, "event_ms_per_do_batch": float(np.mean(ms)), "scan_ms": sc, "stage_ms": stage}

    counted = None
    if nfl > 1 and world == 1 and args.steps >= 2:
        try:   # (a diagnostic leg must never cost the line)
            sb = batch   # alone on the device now: one batch in flight, one unit group — control and scan passes alternate strictly
            sb.run_range(0, B_, sync=True)
            sb.reset_counters()
            serial = serial_steps(sb, [1 % ring, 2 % ring])
            serial["counters"] = sb.counters().astype(np.float64)
            serial["work_counters"] = bool(args.work_counters)
        except Exception as e:
            serial = None
            serial_err = str(e)[:300]
    if not args.work_counters and not args.profile_serial:
        # The counted pass: every resident step once more, alone on the device, with the work counters on.  A step's counters are a pure
        # function of its reads, so the timed region's counters are the sum over its steps of their slots' — the numerator of the SURVEY 8(d)
        # formula — and the same two steps as above give the scan kernel's time WITH the classification (A/B on this box, same minute).
        try:
            batch.set_work_counters(True)
            per_slot = []
            for k in range(ring):
                batch.reset_counters()
                batch.run_range(k * B_, B_, sync=True)
                per_slot.append(batch.counters().astype(np.float64))
            counters = sum(per_slot[i % ring] for i in range(args.warmup, args.warmup + args.steps))
            if serial:
                batch.reset_counters()
                counted = serial_steps(batch, [1 % ring, 2 % ring])
                counted["counters"] = batch.counters().astype(np.float64)
                serial["counters"] = counted["counters"]   # (the same two steps: what the uncounted replay evaluated, with the words counted)
        except Exception as e:
            counted = {"error": str(e)[:300]}
        finally:
            batch.set_work_counters(False)
    reads_per_unit = 2 if pe else 1
    n_reads_rank = args.steps * B_ * reads_per_unit
    # stats reduction: the only collective of the path (RCCL all-gather of a few doubles per rank)
    from bsmap_amd import sharding
    dt_max, tot_counters, allstats = sharding.gather_stats(dt, counters, dist, device="cuda")
    if dist is not None and os.environ.get("BSX_TRACE_COLLECTIVE"):
        print("collective: backend %s world %d all_gather ok (%d rows)" % (dist.get_backend(), world, allstats.shape[0]), file=sys.stderr, flush=True)
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
    sha = lib_sha16()
    traffic, traffic_note, pmc_j = None, "no PMC summary for this build (profiles/pmc_latest.json is from another libbsx.so)", None
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc) and args.mode == "pe":
        try:
            pj = json.load(open(pmc))
            if pj.get("lib_sha16") == sha:
                traffic, traffic_note, pmc_j = pj.get("hbm_bytes_per_launch"), f"profiles/{pj.get('tag')}_pmc.json (same libbsx.so, serial mode)", pj
        except Exception:
            pass
    step_s = k_ms * 1e-3
    wc = bool(args.work_counters)
    dk = dominant_kernel(serial["counters"], serial["scan_ms"], 2, 1, args.mode == "rrbs", args.mode, wc, B_) if serial else dominant_kernel(counters, scan_ms, args.steps, nfl, args.mode == "rrbs", args.mode, wc, B_)
    hbm = None
    if pmc_j:  # what the memory system really moved per step (FETCH_SIZE raw and with the guide's gfx950 x2 rule for wide reads, WRITE_SIZE)
        raw = pmc_j["fetch_bytes_per_step_raw"] + pmc_j["write_bytes_per_step"]
        x2 = pmc_j["fetch_bytes_per_step_x2_gfx950"] + pmc_j["write_bytes_per_step"]
        hbm = {"raw_GBps": raw / step_s / 1e9, "x2_corrected_GBps": x2 / step_s / 1e9, "frac_of_peak_raw": raw / step_s / 1e9 / HBM_PEAK_GBS,
               "frac_of_peak_x2": x2 / step_s / 1e9 / HBM_PEAK_GBS, "bytes_per_step_raw": raw, "bytes_per_step_x2": x2,
               "note": "measured HBM bytes of one step (PMC passes, serial mode) over the wall time of a step in the timed region"}
    bound = "VALU issue and the gathers (texture addresser) of the scan kernel, random 64-byte requests of k_align; NOT HBM bandwidth"
    if hbm:
        bound += " (HBM carries %.2f-%.2f of its peak)" % (hbm["frac_of_peak_raw"], hbm["frac_of_peak_x2"])
    out = {
        "metric": M["metric"], "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic" if not real_fa else "synthetic reads sampled from " + os.path.basename(real_fa),
        "config": {"workload": M["workload"], "pairs_per_step" if pe else "reads_per_step": B_, "genome_bp": int(sum(lens)), "index_entries": int(ref.n_entries),
                   "parallelism": f"read-sharded x{world}",
                   # what the one collective of the path saw: a row per rank of the all-gather (RCCL under the launcher), each rank's own time per step — `value` uses the slowest
                   "ranks": {"seen_by_all_gather": int(allstats.shape[0]), "backend": (dist.get_backend() if dist is not None else None),
                             "ms_per_step_by_rank": [round(float(t) / args.steps * 1e3, 3) for t in allstats[:, 0]], "rank0_pinned_cpus": rank_cpus}, "batches_in_flight": nfl, "work_counters_in_timed_region": bool(args.work_counters), "lib_sha16": sha, "exact_mode": bool(args.exact),
                   "resident_ring_steps": ring, "heavy_pools": [{"units_per_round": u_, "scan_tasks": t_} for u_, t_ in pools], "device_memory_plan": plan,
                   "setup_s": {"genome": round(t_gen, 2), "index_build_gpu": round(t_index, 2), "device_batches": round(t_batches, 2)},
                   "records_flagged_BSX_F_LIMIT": int(tot_counters[16]) if len(tot_counters) > 16 else None,   # (the one capacity deviation from the reference, include/bsx.h: counted, 0)
                   "aligned_fraction": float((2 * tot_counters[6] + tot_counters[5]) / max(1.0, n_reads_rank * world)) if pe
                   else float(tot_counters[5] / max(1.0, n_reads_rank * world))},
        # The roofline of the dominant kernel against the unit that binds it: `achieved` = candidates per second of k_hscan measured live (HIP events
        # on its stream, device counters), `frac` = the utilisation of its busiest unit (counter passes of this kernel under profiles/), `peak` =
        # achieved / frac — the rate at which that unit would be saturated.  The SURVEY 8(d) formula (algorithmic bytes of a step over its
        # wall time against the HBM peak) is kept under `formula_rate`: it is a rate of ALGORITHMIC bytes, not HBM utilisation — the sorted
        # scan order serves most of those bytes from L2 — and can exceed 1.  `hbm_traffic` is what the memory system really moved.
        "roofline": roofline_block(dk, bound, achieved, traffic, traffic_note, hbm, k_ms, kernel_ms, heavy_last, alg_bytes_launch, counters, n_reads_rank, args, pmc_j, serial, np),
    }
    if world == 1 and not args.profile_serial:
        out["workload_shape"] = workload_shape(ref, counters, args.steps, B_, reads_per_unit, heavy_last[0])
    if serial_err:
        out["roofline"]["serial_replay"] = {"error": serial_err}
    if counted:
        if "error" in counted:
            out["roofline"]["with_work_counters"] = counted
        else:
            ck = dominant_kernel(counted["counters"], counted["scan_ms"], 2, 1, args.mode == "rrbs", args.mode, True, B_) or {}
            out["roofline"]["with_work_counters"] = {"serial_ms_per_step": counted["ms_per_step"], "scan_kernel_ms_per_step": ck.get("ms_per_step"), "scan_kernel_candidates_per_s": ck.get("candidates_per_s"),
                                                     "note": "the same two serial steps with bsx_batch_set_work_counters(1): the scan kernels also classify every candidate by the reference's two early-outs "
                                                             "(align.h:189-197) — what the parity suite runs, and where the counters of formula_rate come from; the timed region and dominant_kernel run without"}
    if serial:
        out["roofline"]["serial_replay"] = {"ms_per_step": serial["ms_per_step"], "event_ms_per_do_batch": serial["event_ms_per_do_batch"],
                                            "note": "two steps with one batch in flight and one unit group after the timed region: the source of dominant_kernel"}
    if world == 1 and not args.profile_serial:
        try:  # measured ceilings of this device beside the spec peak
            pm = B.probe_memory(local_rank, 4 << 30, 1 << 30)
            out["roofline"]["peak_measured"] = {"stream_read_GBps": pm["stream_read"], "stream_copy_GBps": pm["stream_copy"],
                                                "gather16_GBps": pm["gather16"], "gather16_Gloads_per_s": pm["gather16_Gloads_per_s"],
                                                "frac_of_stream_read": achieved / pm["stream_read"],
                                                "note": "gather16 = 16-byte loads at random 4-byte-aligned addresses in a 1 GiB window (each moves a 64-byte sector): "
                                                        "the access pattern of the scan; its rate does not improve for windows down to 64 MiB (profiles/r02e_probe_sweep.json)"}
            try:
                fr = fabric_requests(args.mode, wc, B_, out["ms_per_step"], serial["ms_per_step"] if serial else None, pm["gather16_Gloads_per_s"])
                if fr:
                    out["roofline"]["fabric_requests"] = fr
            except Exception as e:
                out["roofline"]["fabric_requests"] = {"error": str(e)[:200]}
        except Exception as e:
            out["roofline"]["peak_measured"] = {"error": str(e)[:200]}
    if world == 1 and args.cpu_seconds > 0:
        try:
            out["cpu_baseline"] = cpu_baseline(ref, batch, pe, kw, args.cpu_seconds, min(args.warmup, ring - 1) * B_, M["kind"] == 1)
        except Exception as e:
            out["cpu_baseline"] = {"error": str(e)[:300]}
    if args.transfer_steps < 0:
        args.transfer_steps = args.steps
    if world == 1 and args.transfer_steps > 0:
        try:
            t_slots = min(ring, args.transfer_steps, args.steps)
            host_reads = [(batch.download_reads(m), batch.download_quals(m) if M["kind"] == 1 else None) for m in range(2 if pe else 1)]
            for bt in batches:   # (the leg's own batches need the room)
                bt.close()
            nt = plan["transfer_leg_batches"]
            vt = incl_transfers(B, ref, host_reads, Align, pe, B_, min(args.transfer_steps, args.steps), nt, reads_per_unit, t_slots, wc)
            del host_reads
            # SURVEY 8(d)'s window is "first batch submitted -> last result returned"; `value` keeps reads and records in HBM (the contract's
            # definition): the same steps with the PCIe legs inside the window, and how far the two are apart
            vt["vs_resident"] = vt["value"] / value
            out["value_incl_transfers"] = vt
        except Exception as e:
            out["value_incl_transfers"] = {"error": str(e)[:300]}
    for bt in batches:
        bt.close()
    ref.close()
    if world == 1 and args.mode == "pe" and args.sensitivity and not real_fa and not args.profile_serial:
        out["sensitivity"] = sensitivity(B, Align, kw, lens, read_len, B_, nfl, M["kind"], value)
    if world == 1 and args.mode == "pe" and args.e2e_pairs > 0:
        out["end_to_end"] = end_to_end(args.e2e_pairs, args.genome)
    if world == 1 and args.mode == "pe" and args.other_configs and not args.profile_serial and not args.exact:
        out["other_configs"] = other_configs(args)
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def roofline_block(dk, bound, achieved, traffic, traffic_note, hbm, k_ms, kernel_ms, heavy_last, alg_bytes_launch, counters, n_reads_rank, args, pmc_j, serial, np):
    """`achieved` = (candidate, read) evaluations per second of the scan kernel, live (HIP events on its stream, device counters).  `peak` = the rate at which the
    kernel would run if it issued NOTHING but its inner word (3 nw + 3 vector instructions per 64 evaluations of nw 32-nt words) at the issue ceiling
    tools/microbench/valu_issue measured for that very word at the kernel's residency — from the microbenchmark, not from the kernel.  `frac` = achieved / peak: the USEFUL
    fraction.  `issue_utilisation` is what round 5 printed as frac: all vector instructions the kernel issues (overhead included: SQ_INSTS_VALU of the same build) over
    the same ceiling; `instructions_per_evaluation` says how many of those it spends per 64 evaluations.  Flat scalars and a short `bound`: the driver's record keeps
    only those (strings up to 120 characters)."""
    dk = dk or {}
    cand_s = dk.get("candidates_per_s")
    ev = dk.get("bound_evidence") if isinstance(dk.get("bound_evidence"), dict) else {}
    peak = ev.get("peak_candidates_per_s")
    stage = (serial or {}).get("stage_ms") or {}
    kern = {"k_align": stage.get("k_align"), "k_hctrl": stage.get("k_hctrl"), dk.get("name") or "scan": stage.get("scan")}
    binding = max((v, k) for k, v in kern.items() if v is not None)[1] if any(v is not None for v in kern.values()) else None
    r = {"bound": short_bound(dk, binding, stage),
         "achieved": cand_s / 1e9 if cand_s else None, "peak": peak / 1e9 if peak else None, "unit": "G candidate-read evaluations/s (" + str(dk.get("name")) + ")",
         "frac": (cand_s / peak) if (cand_s and peak) else None,
         "issue_utilisation": dk.get("binding_unit_utilisation") if dk.get("binding_unit") == "valu_issue" else (ev.get("fractions") or {}).get("valu_issue"),
         "instructions_per_evaluation": ev.get("instructions_per_evaluation"), "inner_word_instructions": ev.get("inner_word_instructions"),
         "binding_kernel": binding, "serial_ms_k_align": stage.get("k_align"), "serial_ms_k_hctrl": stage.get("k_hctrl"), "serial_ms_order": stage.get("order"),
         "serial_ms_scan": stage.get("scan"), "serial_control_passes": stage.get("control_passes"), "serial_ms_per_step": (serial or {}).get("ms_per_step"),
         "bound_long": dk.get("bound") or bound,
         "formula_rate": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                          "note": "SURVEY 8(d): algorithmic bytes of a step / its wall time, against the HBM peak; " + bound},
         "traffic": traffic, "traffic_source": traffic_note, "hbm_traffic": hbm,
         "kernel": "one Do_Batch = k_align + heavy pipeline (k_hctrl/k_hscan passes)", "kernel_ms": k_ms,
         "event_ms_per_do_batch": float(np.mean(kernel_ms)), "heavy_units_last_step": heavy_last[0], "redo_units_last_step": heavy_last[1], "algorithmic_bytes_per_launch": alg_bytes_launch,
         "per_read": {"n_lookup": float(counters[0]) / n_reads_rank, "n_cand": float(counters[1]) / n_reads_rank, "ref_words64": float(counters[2]) / n_reads_rank},
         "group_share": float(counters[15]) / max(1.0, float(counters[7])),   # of the scan kernel's candidates: evaluated in groups of two tasks and more over one window and read offset (k_hscan_same), one fetch and shift for all its reads
         "per_kernel": per_kernel_split(counters, args.steps, n_reads_rank, pmc_j, serial),
         "dominant_kernel": dk}
    return r


def short_bound(dk, binding, stage):
    """<= 120 characters (the driver's record cuts strings there): the step's binding kernel, and what the scan kernel is bound by"""
    ev = dk.get("bound_evidence") if isinstance(dk.get("bound_evidence"), dict) else {}
    fr = ev.get("fractions") or {}
    unit = {"valu_issue": "VALU issue", "texture_addresser_busy": "texture path", "lds_active": "LDS"}.get(dk.get("binding_unit"), "no counter file")
    txt = "%s: %s %.2f of own ceiling" % (dk.get("name"), unit, dk.get("binding_unit_utilisation") or 0.0) if dk.get("binding_unit") else "%s: no counter file of this build" % dk.get("name")
    if fr.get("l2_hit") is not None:
        txt += ", L2 hit %.2f, not HBM" % fr["l2_hit"]
    if binding and stage.get("scan"):
        txt += "; serial step: %s longest" % binding
    return txt[:120]


def workload_shape(ref, c, steps, units_per_step, reads_per_unit, heavy_units):
    """what the headline depends on: the bucket-size distribution of the seed index (3-letter buckets of repeats hold millions of entries)
    and how the candidates split between the units the main kernel finishes itself and the few it hands to the heavy pipeline"""
    import numpy as np
    out = {}
    try:
        off = ref.index()[0].astype(np.int64)
        sizes = np.diff(off)
        ne = sizes[sizes > 0]
        tot = float(sizes.sum())
        out["index_buckets"] = {"buckets": int(len(sizes)), "non_empty": int(len(ne)), "entries": int(tot), "mean_non_empty": float(ne.mean()), "median_non_empty": float(np.median(ne)),
                                "p99_non_empty": float(np.percentile(ne, 99)), "p99.99_non_empty": float(np.percentile(ne, 99.99)), "max": int(sizes.max()),
                                "share_of_entries_in_buckets_over_64k": float(sizes[sizes > 65536].sum() / tot), "buckets_over_64k": int((sizes > 65536).sum()),
                                "share_of_entries_in_buckets_over_1k": float(sizes[sizes > 1024].sum() / tot)}
        del off, sizes, ne
    except Exception as e:
        out["index_buckets"] = {"error": str(e)[:200]}
    units = float(steps * units_per_step)
    hv = float(heavy_units) * steps   # (units deferred in the last step x steps: every step aligns different reads of the same distribution)
    main_c, all_c = float(c[12]), float(c[1])
    out["candidates_by_unit_class"] = {
        "finished_by_the_main_kernel": {"share_of_units": (units - hv) / units, "candidates_per_read": main_c / max(1.0, (units - hv) * reads_per_unit), "share_of_candidates": main_c / max(1.0, all_c)},
        "deferred_to_the_heavy_pipeline": {"share_of_units": hv / units, "candidates_per_read": (all_c - main_c) / max(1.0, hv * reads_per_unit), "share_of_candidates": (all_c - main_c) / max(1.0, all_c),
                                           "note": "units with a seed in a repeat bucket: a candidate list of 32 768 or more (RRBS 2 048)"}}
    return out


def per_kernel_split(c, steps, n_reads, pmc_j, serial):
    """algorithmic bytes per step by kernel, from the device counters alone (SURVEY 8(d) terms): counters 11-14 are the share of
    0-3 the main kernel did itself, 7-8 what the scan kernel evaluated (incl. the little it evaluates speculatively); the rest of
    the heavy units' work is the control kernel's inline scans.  With the PMC summary of the same build: measured HBM bytes per
    kernel and their ratio to the algorithmic bytes (the amplification of 64-byte sectors under 8-/16-byte gathers)."""
    c = [float(x) / steps for x in c]
    main = {"n_lookup": c[11], "n_cand": c[12], "ref_words64": c[13], "n_orient": c[14]}
    main["algorithmic_bytes"] = 8 * c[11] + 4 * c[12] + 8 * c[13] + 80 * c[14] + 16 * n_reads / steps
    heavy = {"n_lookup": c[0] - c[11], "n_cand": c[1] - c[12], "ref_words64": c[2] - c[13], "n_orient": c[3] - c[14]}
    heavy["algorithmic_bytes"] = 8 * heavy["n_lookup"] + 4 * heavy["n_cand"] + 8 * heavy["ref_words64"] + 80 * heavy["n_orient"]
    scan = {"n_cand": c[7], "ref_words64": c[8], "algorithmic_bytes": 4 * c[7] + 8 * c[8],
            "note": "includes windows evaluated speculatively and later re-published; a subset of heavy_units' candidates otherwise"}
    out = {"k_align": main, "heavy_units_total(k_hctrl+k_hscan)": heavy, "k_hscan": scan}
    if pmc_j:
        for name, key in (("k_align", "k_align"), ("k_hscan", "k_hscan"), ("k_hctrl", "k_hctrl")):
            e = [v for k, v in pmc_j["kernels"].items() if key in k]
            if not e:
                continue
            e = e[0]
            raw = e.get("fetch_bytes_per_launch_raw", 0) * e["launches_per_step"] + e.get("write_bytes_per_launch", 0) * e["launches_per_step"]
            d = out.setdefault(name, {})
            d["hbm_bytes_per_step_raw"] = raw
            d["hbm_bytes_per_step_x2"] = raw + e.get("fetch_bytes_per_launch_raw", 0) * e["launches_per_step"]
            if d.get("algorithmic_bytes"):
                d["hbm_over_algorithmic_raw"] = raw / d["algorithmic_bytes"]
    return out


def dominant_kernel(counters, scan_ms, steps, nfl, rrbs=False, mode="pe", work_counters=False, units_per_step=None):
    """The scan kernel of the heavy pipeline (k_hscan_same; RRBS k_hscan_shared), the kernel most of the time goes to: launches and HIP-event durations measured live (events on the stream it
    is launched on), algorithmic bytes of the candidates it evaluated (4 B index entry + 8 B per 64-bit reference word the
    reference's CountMismatch would touch).  `bound` is derived from the committed counter summaries of the
    same kernel, build, mode and counter setting (VALU issue rate against the ceiling of ITS instruction mix measured by
    tools/microbench/valu_issue at its own residency, texture-addresser busy fraction), not asserted: see profiles/README.md."""
    tot_ms = float(sum(t for t, n in scan_ms)); launches = int(sum(n for t, n in scan_ms))
    cand, words = float(counters[7]), float(counters[8])
    alg = 4.0 * cand + 8.0 * words
    if launches == 0 or tot_ms <= 0:
        return None
    same_env = os.environ.get("BSX_SAME", "1")
    d = {"name": ("k_hscan_same" if same_env == "2" else "k_hscan_shared") if rrbs else ("k_hscan" if same_env == "0" else "k_hscan_same"), "launches_per_step": launches / steps, "avg_launch_ms": tot_ms / launches, "ms_per_step": tot_ms / steps,
         "candidates_per_launch": cand / launches, "algorithmic_bytes_per_launch": alg / launches,
         "achieved_GBps": alg / (tot_ms * 1e-3) / 1e9, "frac_of_hbm_peak": alg / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
         "candidates_per_s": cand / (tot_ms * 1e-3),
         "class_shares": {"one_word": float(counters[9]) / max(cand, 1.0), "five_words": float(counters[10]) / max(cand, 1.0)},
         "timing_note": "launch durations overlap other kernels when batches_in_flight > 1" if nfl > 1 else "serial: no other kernel runs beside it"}
    d.update(kernel_bound(mode, work_counters, cand / steps, cand / (tot_ms * 1e-3), units_per_step))
    return d


def profile_summary(kind, mode, work_counters):
    """the newest (by name = round tag) profiles/r*_<kind>*.json taken with THIS libbsx.so in this mode and counter setting, or None.  Files of
    rounds 1-4 carry no mode: *_rrbs.json is RRBS, the others C3, all counted."""
    import glob
    sha = lib_sha16()
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s*.json" % kind))):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        if j.get("lib_sha16") != sha or "kernels" not in j:
            continue
        if j.get("mode", "rrbs" if f.endswith("_rrbs.json") else "pe") != mode or bool(j.get("work_counters", 1)) != bool(work_counters):
            continue
        if j.get("scan_kernel", "1") != os.environ.get("BSX_SAME", "1"):
            continue
        best = (f, j)
    return best


def valu_ceiling(mix_prefix, waves):
    """wave64 vector instructions per second the chip sustains for an instruction mix at `waves` resident waves per SIMD (tools/microbench/valu_issue
    measured 1 / 2 / 4 / 6 / 8; linear in between), from the newest profiles/r*_valu_issue.json that holds the mix"""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu_issue.json")), reverse=True):
        vi = json.load(open(f))
        mx = [m for m in vi["mixes"] if m["mix"].startswith(mix_prefix)]
        if not mx:
            continue
        pts = sorted((int(w), v["chip_G_wave_instr_per_s"] * 1e9) for w, v in mx[0]["by_waves_per_simd"].items())
        w = min(max(waves, pts[0][0]), pts[-1][0])
        for (w0, r0), (w1, r1) in zip(pts, pts[1:]):
            if w0 <= w <= w1:
                return r0 + (r1 - r0) * (w - w0) / (w1 - w0), pts[-1][1], os.path.basename(f), mx[0]["mix"]
    return None


INNER_WORDS = {"pe": 5, "trim": 5, "se": 4, "rrbs": 3}   # 32-nt words of a read in the scan kernels (144 / 100 / 75 nt)


def inner_word_instructions(mode, work_counters):
    """vector instructions of the scan kernels' inner word per 64 (candidate, read) evaluations: per 32-nt word two v_bitop3 (three for the last) and one v_bcnt,
    the candidate's share of the funnel shifts and the compare — 3 nw + 3 (nw = 5: tools/microbench/valu_issue mix 16, 18 instructions); with the work counters
    two more v_bcnt, two v_and and two compares (mix 17, 24)"""
    return 3 * INNER_WORDS[mode] + 3 + (6 if work_counters else 0)


def kernel_bound(mode="pe", work_counters=False, cand_per_step=None, cand_per_s=None, units_per_step=None):
    """utilisation of the scan kernel's units from measurements kept under profiles/: SQ / TA / LDS counter passes of the kernel (tools/sq_passes.sh,
    tools/summarize_sq.py) taken with this library, mode and counter setting — nothing borrowed from another build or config: without such a file the
    fractions are null — and the VALU issue ceiling of the kernel's OWN inner word (tools/microbench/valu_issue.hip, mixes 16 / 17) at the kernel's own
    residency.  The binding unit is the busiest one."""
    none = {"bound": None, "binding_unit": None, "binding_unit_utilisation": None,
            "bound_evidence": "no counter summary of this build (lib_sha16 %s), mode %s, work counters %s under profiles/" % (lib_sha16(), mode, "on" if work_counters else "off")}
    try:
        got = profile_summary("sq", mode, work_counters)
        if not got:
            return none
        sq, j = got
        k = [v for n, v in j["kernels"].items() if n.startswith("k_hscan")][0]["derived"]
        waves = k.get("resident_waves_per_simd") or 4.0
        vc = valu_ceiling("k_hscan_same inner word with work counters" if work_counters else "k_hscan_same inner word (", waves)
        if not vc:
            return none
        ceil, ceil_full, vif, mixname = vc
        inner = inner_word_instructions(mode, work_counters)
        kk = [v for n, v in j["kernels"].items() if n.startswith("k_hscan")][0]
        ipe = None
        if cand_per_step and kk["counters"].get("SQ_INSTS_VALU"):   # vector instructions per 64 evaluations: the profile's count per step over the live candidates of a step
            scale = (units_per_step / float(j["units_per_step"])) if (units_per_step and j.get("units_per_step")) else 1.0
            ipe = kk["counters"]["SQ_INSTS_VALU"] / float(j.get("steps_in_pass") or 3) * scale / (cand_per_step / 64.0)
        fr = {"valu_issue": k["valu_instr_per_s"] / ceil, "valu_issue_at_full_occupancy": k["valu_instr_per_s"] / ceil_full, "texture_addresser_busy": k.get("ta_busy_frac"), "lds_active": k.get("lds_active_frac"),
              "l2_hit": k.get("l2_hit_frac"), "l1_miss_per_access": k.get("l1_miss_per_access"), "waiting_on_instruction_issue": k.get("wait_inst_frac"), "waiting_for_data": k.get("wait_any_frac"),
              "resident_waves_per_simd": waves}
        top = max((v, n) for n, v in fr.items() if v is not None and n in ("valu_issue", "texture_addresser_busy", "lds_active"))
        names = {"valu_issue": "VALU issue (against the ceiling of the kernel's own inner word at its %.1f resident waves per SIMD)" % waves,
                 "texture_addresser_busy": "texture path (64-lane gathers: the addresser's busy time)", "lds_active": "LDS"}
        return {"bound": f"{names[top[1]]} at {top[0]:.2f}; the others: " + ", ".join(f"{names[n].split(' (')[0]} {fr[n]:.2f}" for n in names if n != top[1] and fr[n] is not None) + f"; not HBM (L2 hit {fr['l2_hit']:.2f})",
                "binding_unit": top[1], "binding_unit_utilisation": top[0],
                "bound_evidence": {"fractions": fr, "valu_ceiling_G_wave_instr_per_s": ceil / 1e9, "valu_ceiling_at_8_waves_per_simd": ceil_full / 1e9, "valu_mix": mixname, "same_build": True,
                                   "inner_word_instructions": inner, "instructions_per_evaluation": ipe, "peak_candidates_per_s": ceil / inner * 64.0,
                                   "useful_frac": (cand_per_s / 64.0 * inner / ceil) if cand_per_s else None,
                                   "sources": [os.path.basename(sq), vif]}}
    except Exception as e:
        none["bound_evidence"] += " (%s)" % str(e)[:120]
        return none


def fabric_requests(mode, work_counters, B_, ms_per_step, serial_ms, gather_rate):
    """The bound of the whole step: random 64-byte read requests to the fabric.  Per kernel TCC_EA0_RDREQ per step from the counter
    passes of this build / mode / counter setting (serial mode), the live rate of random 16-byte gathers on this box, and how much of the step the
    requests alone would take at that rate."""
    got = profile_summary("sq", mode, work_counters)
    if not got or not gather_rate:
        return None
    f, j = got
    scale = B_ / float(j.get("units_per_step") or B_)
    per = {n: v["derived"]["fabric_read_requests_per_step"] * scale for n, v in j["kernels"].items() if "fabric_read_requests_per_step" in v["derived"]}
    tot = sum(per.values())
    t_ms = tot / (gather_rate * 1e9) * 1e3
    out = {"requests_per_step": tot, "by_kernel": per, "gather16_Gloads_per_s": gather_rate, "ms_at_that_rate": t_ms, "frac": t_ms / ms_per_step,
           "frac_of_serial_step": (t_ms / serial_ms) if serial_ms else None, "source": os.path.basename(f),
           "note": "TCC_EA0_RDREQ (64-byte read requests that left an L2) per step, summed over the step's kernels, over the rate of random 16-byte gathers "
                   "measured live on this box (peak_measured.gather16): the time the memory system needs for the step's requests alone, as a fraction of the step"}
    if scale != 1.0:
        out["note"] += "; counters taken at %d units per step, scaled to %d" % (j.get("units_per_step"), B_)
    return out


def sensitivity(B, Align, kw, lens, read_len, B_, nfl, kind, headline):
    """the same measurement (3 timed steps, inputs resident) on variants of the synthetic genome: 96 % of the headline's candidates come
    from the 2 % of pairs that fall into microsatellite / poly-A classes, so the number moves with the generator's repeat content"""
    import threading
    import numpy as np
    out = {"headline_reads_per_s": headline, "variants": {}}
    for name, var in (("microsatellite_windows_halved", 1), ("no_repeat_elements", 2)):
        ref = B.RefSeq(B.make_params(**kw)).synthetic(lens, seed=38 | (var << 56)).CreateIndex()
        steps, warm = (4, 2) if nfl == 2 else (3, 1)   # (every batch in flight runs the same number of steps)
        bts = [Align(ref, B_ * (steps + warm)).set_work_counters(False) for _ in range(nfl)]   # (as the timed region; candidates_per_read then misses the few the count-only walks would add)
        for bt in bts:
            bt.synth_reads(B_ * (steps + warm), read_len, seed=3, kind=kind)

        def run(lo, hi):
            def w(j):
                for i in range(lo + j, hi, nfl):
                    bts[j].run_range(i * B_, B_, sync=True)
            th = [threading.Thread(target=w, args=(j,)) for j in range(1, nfl)]
            for t in th:
                t.start()
            w(0)
            for t in th:
                t.join()
        run(0, warm)
        for bt in bts:
            bt.reset_counters()
        t0 = time.perf_counter()
        run(warm, warm + steps)
        dt = time.perf_counter() - t0
        c = sum(bt.counters().astype(np.float64) for bt in bts)
        n_reads = steps * B_ * (2 if kw.get("pairend") else 1)
        out["variants"][name] = {"reads_per_s": n_reads / dt, "ms_per_step": dt / steps * 1e3, "candidates_per_read": float(c[1]) / n_reads,
                                 "lookups_per_read": float(c[0]) / n_reads, "heavy_units_last_step": int(bts[0].heavy_units())}
        for bt in bts:
            bt.close()
        ref.close()
    return out


def other_configs(args):
    """BASELINE.json's other single-GPU configs, each in a child process of its own (this process has released its device memory): the same
    timed region as the metric's (inputs in HBM, barrier + synchronize on both sides), fewer steps, none of the side legs.  A child that fails
    is reported with its return code and the end of its stderr, and tried once more with two batches in flight."""
    import subprocess
    res = {}
    for mode in ("se", "rrbs", "trim"):
        base = [sys.executable, os.path.abspath(__file__), "--mode", mode, "--steps", "6", "--warmup", "3", "--genome", str(args.genome),
                "--cpu-seconds", "0", "--transfer-steps", "0", "--e2e-pairs", "0", "--other-configs", "0"]
        if getattr(args, "units_given", False):   # (otherwise every mode runs its own default step size)
            base += ["--pairs-per-step", str(args.pairs_per_step)]
        tag, failures = MODES[mode]["tag"], []
        for extra in ([], ["--in-flight", "2"]):
            t0 = time.perf_counter()
            try:
                r = subprocess.run(base + extra, capture_output=True, text=True, timeout=600)
                lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if r.returncode != 0 or not lines:
                    failures.append({"args": " ".join(extra), "rc": r.returncode, "stderr_tail": r.stderr[-300:]})
                    continue
                j = json.loads(lines[-1])
                dk = j["roofline"].get("dominant_kernel") or {}
                upu = j["config"].get("pairs_per_step") or j["config"].get("reads_per_step")
                res[tag] = {"workload": j["config"]["workload"], "reads_per_s": j["value"], "ms_per_step": j["ms_per_step"], "units_per_step": upu,
                            "ms_per_2^20_units": j["ms_per_step"] * (1 << 20) / upu if upu else None, "steps": j["steps"],
                            "batches_in_flight": j["config"]["batches_in_flight"], "aligned_fraction": j["config"]["aligned_fraction"], "heavy_pools": j["config"].get("heavy_pools"),
                            "candidates_per_read": j["roofline"]["per_read"]["n_cand"], "dominant_kernel": {k: dk.get(k) for k in ("name", "ms_per_step", "candidates_per_s")},
                            # one serial step by kernel (HIP events, one batch in flight: they add up), which of them is the longest, and the scan kernel's fractions
                            "serial_ms": {k: j["roofline"].get("serial_ms_" + k) for k in ("k_align", "k_hctrl", "order", "scan", "per_step")}, "binding_kernel": j["roofline"].get("binding_kernel"),
                            "control_passes": j["roofline"].get("serial_control_passes"),
                            "scan_kernel_frac": j["roofline"].get("frac"), "scan_kernel_issue_utilisation": j["roofline"].get("issue_utilisation"),
                            "scan_kernel_instructions_per_evaluation": j["roofline"].get("instructions_per_evaluation"), "wall_s": round(time.perf_counter() - t0, 1)}
                if failures:
                    res[tag]["failed_attempts"] = failures
                break
            except Exception as e:
                failures.append({"args": " ".join(extra), "error": str(e)[:300]})
        if tag not in res:
            res[tag] = {"error": "no result line", "attempts": failures}
    return res


def pinned_array(B, C, nbytes):
    import numpy as np
    p = B.lib().bsx_pinned_alloc(nbytes)
    if not p:
        return np.zeros(nbytes, np.uint8), None
    return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(p)), p


def incl_transfers(B, ref, host_reads, Align, pe, B_, steps, nt, reads_per_unit, slots, work_counters=False):
    """the second headline: the same Do_Batch with the PCIe legs inside the timed window — per step the reads of that step go up
    from page-locked host memory (bsx_batch_upload_*), the batch runs, and the records come back (bsx_batch_results_*);
    `nt` device batches on as many host threads as in the command-line driver, so one batch's transfers overlap the other's kernels.  The host side
    holds `slots` distinct steps of the resident run (host_reads: per mate ((bytes, offsets), quals or None); step i uploads slot i % slots);
    everything allocated here is released on any exit."""
    import ctypes as C
    import threading
    import numpy as np
    L = B.lib()
    L.bsx_pinned_alloc.restype = C.c_void_p
    L.bsx_pinned_alloc.argtypes = [C.c_size_t]
    L.bsx_pinned_free.argtypes = [C.c_void_p]
    host, pins, small = [], [], []
    try:
        for (buf, off), qsrc in host_reads:
            hi = int(off[slots * B_])
            a, p = pinned_array(B, C, hi)
            pins.append(p)
            a[:] = buf[:hi]
            q = None
            if qsrc is not None:
                q, pq = pinned_array(B, C, hi)
                pins.append(pq)
                q[:] = qsrc[:hi]
            host.append((a, off[:slots * B_ + 1].astype(np.uint64), q))
        nt = int(os.environ.get("BSX_T_BATCHES", nt))            # (experiments: batches of this leg, and how many of them may be inside Do_Batch at once)
        gate = threading.Semaphore(int(os.environ.get("BSX_T_GATE", nt)))
        for _ in range(nt):
            small.append(Align(ref, B_).set_work_counters(work_counters))
        sinks = []    # page-locked result arrays per batch
        for _ in range(nt):
            arrs = []
            for dt in ((B.PAIR_DTYPE, B.CC_DTYPE, B.CC_DTYPE) if pe else (B.HIT_DTYPE, B.CC_DTYPE)):
                raw, p_ = pinned_array(B, C, B_ * dt.itemsize)
                pins.append(p_)
                arrs.append(raw.view(dt))
            sinks.append(tuple(arrs))

        # per slot the arrays a caller hands over: bytes of the step's reads and their offsets from 0 (prepared before the window opens — packing the
        # reads of a batch is the caller's parser, which `end_to_end` measures; the window is upload -> Do_Batch -> results)
        step_off = []
        for a, off, q in host:
            per = []
            for i in range(slots):
                raw, p_ = pinned_array(B, C, (B_ + 1) * 8)   # (page-locked like the read bytes: a pageable source is staged through a bounce buffer)
                pins.append(p_)
                o = raw.view(np.uint64)
                o[:] = off[i * B_:(i + 1) * B_ + 1] - off[i * B_]
                per.append(o)
            step_off.append(per)

        def step(j, i):
            b = small[j]
            k = i % slots
            sl = []
            for m_, (a, off, q) in enumerate(host):
                o = step_off[m_][k]
                s0, s1 = int(off[k * B_]), int(off[(k + 1) * B_])
                sl.append((a[s0:s1], o, q[s0:s1] if q is not None else None))
            if pe:
                b.ImportBatchReads((sl[0][0], sl[0][1]), (sl[1][0], sl[1][1]), sl[0][2], sl[1][2], first_index=k * B_)
            else:
                b.ImportBatchReads((sl[0][0], sl[0][1]), sl[0][2], first_index=k * B_)
            with gate:
                b.Do_Batch()
            b.results(into=sinks[j])

        nxt, lock, errs = [0], threading.Lock(), []

        def worker(j, lo, hi):  # a batch takes the next step that nobody has started (a fixed stride would leave batches idle when nt does not divide the steps)
            try:
                while not errs:
                    with lock:
                        i = nxt[0]; nxt[0] += 1
                    if i >= hi:
                        return
                    step(j, i)
            except Exception as e:
                errs.append(e)
        for j in range(nt):  # untimed warm-up of each small batch
            step(j, 0)
        t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(j, 0, steps)) for j in range(1, nt)]
        for t in th:
            t.start()
        worker(0, 0, steps)
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        if errs:
            raise errs[0]
        pool = [b.pool_sizes() for b in small]
        up = sum((int(h[1][-1]) * (2 if h[2] is not None else 1) + 8 * (slots * B_ + 1)) / slots for h in host)
        down = B_ * ((64 + 128) if pe else (16 + 64))
        return {"value": steps * B_ * reads_per_unit / dt, "unit": "reads/s", "steps": steps, "ms_per_step": dt / steps * 1e3,
                "host_to_device_bytes_per_step": up, "device_to_host_bytes_per_step": down, "heavy_pools": [{"units_per_round": u_, "scan_tasks": t_} for u_, t_ in pool],
                "window": "per step: bsx_batch_upload (page-locked read bytes, per-step offset arrays prepared beforehand) -> Do_Batch -> bsx_batch_results (page-locked), %d batches in flight" % nt}
    finally:
        for b in small:
            b.close()
        host.clear()
        for p in pins:
            if p:
                L.bsx_pinned_free(p)


def end_to_end(pairs, genome):
    """the same workload through the command-line driver — FASTA genome + two FASTQ files in, SAM out, all
    in /dev/shm — so that parsing, PCIe, formatting and writing are inside the time.  Reported beside `value`, never as it."""
    import types
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    try:
        import e2e_bench
        # the same input files twice: ONE pipeline (one reader per file, one ring, one output file: its host side tops out near 20 M reads/s on the box's 16 CPUs,
        # DESIGN.md §8), and two lanes that share the GPU (`--lanes=2 --lane-files`: two processes, each with its own reader, ring, replica and output file — the
        # reference's -B / -E shards on one node), the mode to use where one pipeline's host side is the limit
        rr = e2e_bench.run(types.SimpleNamespace(pairs=pairs, genome=1.0 if genome == "hg38" else float(genome), dir="/dev/shm/bsx_e2e_%d" % os.getpid(),
                                                 threads=0, keep=False, variants=";--lanes=2 --lane-files"))
        rs = rr.get("variants") or [rr]

        def leg(r, cmd):
            t = r["timing"]
            return {"reads_per_s": r["reads_per_s_mapping_phase"], "mapping_s": t["mapping_s"], "load_reference_s": t.get("load_reference_s"), "index_build_s": t.get("index_build_s"),
                    "whole_process_s": r["cli_wall_s"], "mapping_cpu_s": t.get("mapping_cpu_s"), "stage_busy_s": t.get("stage_busy_s"), "host_threads": t.get("workers"), "command": cmd}
        base = "bsmap -a r_1.fq -b r_2.fq -d genome.fa -o out.sam -s 16 -v 6 -m 28 -x 500 -S 1"
        out = leg(rs[0], base)
        out.update({"unit": "reads/s, first batch parsed -> last SAM line written", "pairs": pairs, "fastq_bytes": rs[0]["fastq_bytes"], "sam_bytes": rs[0]["sam_bytes"]})
        if len(rs) > 1:
            out["two_lanes_one_gpu"] = leg(rs[1], base + " --lanes=2 --lane-files")
            out["best_reads_per_s"] = max(out["reads_per_s"], out["two_lanes_one_gpu"]["reads_per_s"])
        return out
    except Exception as e:  # the metric above does not depend on this leg
        return {"error": str(e)[:300]}


def rank_cpu_share(cpus_of_node, ranks_on_node, k, per_rank):
    """the CPUs rank k (of `ranks_on_node` ranks that share a NUMA node) pins to: `per_rank` consecutive CPUs of the node's list, the ranks side by side;
    None when the node has fewer than that to give (no pinning then)"""
    cpus = sorted(cpus_of_node)
    if per_rank < 1 or len(cpus) < ranks_on_node * per_rank:
        return None
    return set(cpus[k * per_rank:(k + 1) * per_rank])


def pin_rank_to_gpu_node(B, local_rank, world):
    """A rank's host side is its driver threads (one per batch in flight; they sleep on events) and the Python that queues them: pin the process to its
    share of the CPU quota on the NUMA node of ITS GPU, so that page-locked buffers, launch queues and the threads that touch them sit beside the device
    (the command line does the same per lane: csrc/bsx_cpus.h).  Returns the CPU set, or None where nothing was pinned (no quota headroom, no sysfs)."""
    try:
        n_dev = max(1, B.lib().bsx_device_count())
        nodes = [B.lib().bsx_device_numa_node(d) for d in range(min(world, n_dev))]
        node = nodes[local_rank] if local_rank < len(nodes) else -1
        if node < 0:
            return None
        have = os.sched_getaffinity(0)
        cpus = []
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus += [c for c in range(int(a), int(b or a) + 1) if c in have]
        same = [r for r in range(len(nodes)) if nodes[r] == node]
        per_rank = max(2, usable_cpus() // max(1, world))
        share = rank_cpu_share(cpus, len(same), same.index(local_rank), per_rank)
        if share:
            os.sched_setaffinity(0, share)
        return sorted(share) if share else None
    except (OSError, ValueError, AttributeError):
        return None


def usable_cpus():
    """CPUs this process can use: the affinity mask cut down to the cgroup CPU quota (a box that shows 256 hardware threads
    may allow 16 CPUs' worth of time; more threads than that only get throttled).  Same rule as csrc/bsx_cpus.h."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = period = None
    try:
        q, p_ = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota, period = int(q), int(p_)
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except (OSError, ValueError):
            pass
    if quota and period and quota > 0:
        n = min(n, max(1, -(-quota // period)))
    return max(1, n)


def node_cpus(node, n):
    """the first n CPUs of NUMA node `node` this process may run on (the same choice as csrc/bsx_cpus.h's bsx_pin_to_node)"""
    try:
        have = os.sched_getaffinity(0)
        out = []
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            out += [c for c in range(int(a), int(b or a) + 1) if c in have]
        return set(out[:n]) if len(out) >= n and len(have) > n else None
    except (OSError, ValueError):
        return None


def cpu_baseline(ref, batch, pe, kw, target_s, first_unit, quals):
    """the oracle (plain-C port of the reference algorithm, pthread batch model of main.cpp:49-73) timed on this box's
    host cores over a bounded sample of the SAME reads against the SAME reference + index (copied back from HBM; RRBS: the
    port packs and indexes the genome text itself first, untimed, like the index build of the GPU path).  Two thread
    counts: every CPU the process may use (usable_cpus()), and 8 — the reference caps its default -p at 8 (param.cpp:8-9)."""
    import numpy as np
    from oracle import oracle_ffi as O
    cores, hw = usable_cpus(), os.cpu_count() or 1
    # Under a CPU quota far below the visible CPUs the scheduler spreads the threads over both sockets; pinned to the quota's worth of
    # CPUs on one NUMA node — before the reference and index are copied back, so that their pages land on that node too — the port
    # runs 1.3-1.4 x faster (35.9 K vs 28.0 K reads/s): the CPU gets that, too.
    import bsmap_amd as B_
    old_aff = os.sched_getaffinity(0)
    pin = node_cpus(max(0, B_.lib().bsx_device_numa_node(0)), cores)
    if pin:
        os.sched_setaffinity(0, pin)
    if kw.get("D"):   # RRBS: site tables and the {tag, loc} index are the oracle's own, built from the genome text pulled back from HBM
        parts = []
        for c, nm in enumerate(ref.names()):
            parts += [np.frombuffer(f">{nm}\n".encode(), np.uint8), ref.synth_bytes(c), np.frombuffer(b"\n", np.uint8)]
        text = np.concatenate(parts).tobytes()
        del parts
        oref = O.OracleRef(O.make_params(**kw), fasta_text=text)
        del text
    else:
        f, c = ref.words()
        a, s, r = ref.info()
        off, nf, ent = ref.index()
        oref = O.OracleRef.wrap(O.make_params(**kw), f, c, a, s, r, off, nf, ent)
    b1, o1 = batch.download_reads(0)
    q1 = batch.download_quals(0) if quals else None
    L = int(o1[1] - o1[0])
    if pe:
        b2, o2 = batch.download_reads(1)
        q2 = batch.download_quals(1) if quals else None

    def run(n, threads):
        lo = first_unit
        oa = (o1[lo:lo + n + 1] - o1[lo]).copy()
        sa = b1[int(o1[lo]):int(o1[lo + n])].copy()
        qa = q1[int(o1[lo]):int(o1[lo + n])].copy() if quals else None
        if pe:
            sb = b2[int(o2[lo]):int(o2[lo + n])].copy()
            ob = (o2[lo:lo + n + 1] - o2[lo]).copy()
            qb = q2[int(o2[lo]):int(o2[lo + n])].copy() if quals else None
            t0 = time.perf_counter()
            O.pe_batch(oref, sa, oa, sb, ob, qa, qb, first_index=lo, threads=threads)
        else:
            t0 = time.perf_counter()
            O.se_batch(oref, sa, oa, qa, first_index=lo, threads=threads)
        return time.perf_counter() - t0

    def sample(threads):
        n0 = min(2000 if threads <= 8 else 20000, len(o1) - 1 - first_unit)
        t_probe = run(n0, threads)
        n = int(min(len(o1) - 1 - first_unit, max(n0, n0 * target_s / max(t_probe, 1e-3))))
        t = run(n, threads)
        return n, t
    try:
        n, t = sample(cores)
        n8, t8 = sample(min(8, cores))
    finally:
        os.sched_setaffinity(0, old_aff)
    rp = 2 if pe else 1
    por, por_src = None, None
    try:   # what the real `bsmap -p 8` does against the port in this regime (build container, tools/cpu_port_vs_reference.py --pe): the port is the slower one
        pj = json.load(open(os.path.join(ROOT, "profiles", "r05_cpu_port_vs_reference_pe.json")))
        por, por_src = float(pj["port_over_reference"]), "profiles/r05_cpu_port_vs_reference_pe.json"
    except Exception:
        pass
    return {"value": n * rp / t, "unit": "reads/s", "cores": cores, "kind": "port", "hardware_threads": hw,
            "port_over_reference": por, "value_reference_equivalent": (n * rp / t / por) if por else None, "port_over_reference_source": por_src,
            "sample": f"{n} {'pairs' if pe else 'reads'} of the timed workload ({L} nt), oracle/bsx_oracle.c with {cores} pthreads"
                      f" (the CPUs this process may use: affinity mask and cgroup quota; the box shows {hw} hardware threads"
                      f"{'; pinned to that many CPUs of one NUMA node' if pin else ''}), {t:.1f} s",
            "p8": {"value": n8 * rp / t8, "cores": min(8, cores), "sample": f"{n8} {'pairs' if pe else 'reads'}, {min(8, cores)} pthreads (the reference's default -p cap), {t8:.1f} s"}}


if __name__ == "__main__":
    main()
