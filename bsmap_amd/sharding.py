"""Read sharding and the final statistics reduction for multi-GPU runs.

Reads (pairs) are independent units (the pick RNG is a pure function of the read index, reference utilities.cpp:44),
so rank r of W simply takes the r-th contiguous block of every batch; reference + index are replicated per GPU.
The only exchange is the end-of-run reduction of the counters the reference keeps per thread and sums under
mutex_fout (n_aligned / n_aligned_pairs / n_aligned_a / n_aligned_b, main.cpp:39-42,70-72) plus the work counters:
one all-gather of a few numbers per rank (RCCL on GPUs, gloo in the CPU tests)."""
import numpy as np


def shard_range(n_units, rank, world):
    """contiguous block [lo, hi) of rank `rank`; blocks differ by at most one unit and cover [0, n_units) exactly"""
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_stats(elapsed_s, counters, dist=None, device="cpu"):
    """all-gather (elapsed, counters...) of every rank; returns (max elapsed over ranks, summed counters, per-rank table)"""
    import torch
    v = torch.tensor([float(elapsed_s)] + [float(x) for x in counters], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized():   # (also at world size 1: the collective is the path — a launcher-started single rank runs it, too)
        parts = [torch.zeros_like(v) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, v)
        table = torch.stack(parts).cpu().numpy()
    else:
        table = v.cpu().numpy()[None, :]
    return float(table[:, 0].max()), table[:, 1:].sum(0), table


def whole_job_rate(units_per_rank_list, reads_per_unit, max_elapsed_s):
    """value of the bench line: all reads of all ranks over the slowest rank's time"""
    return float(np.sum(units_per_rank_list)) * reads_per_unit / max_elapsed_s
