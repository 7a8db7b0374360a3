"""Multi-GPU: sequences shard across ranks (one process per GPU); the only exchange is ONE all-reduce
(SUM) of [sum_seq mean*count ..., count] per (dataset, metric) -- exactly what MetricTracker.update
accumulates (eval.py:259-266).  Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests."""
import torch


import os


def force_collectives():
    """EVR_FORCE_DIST=1: issue the collectives on a one-rank process group too -- how the RCCL path is exercised on a one-GPU box
    (tests/test_gpu_dist.py, bench.py: `rccl_ranks: 1`); a sum over one rank is the identity."""
    return bool(os.environ.get('EVR_FORCE_DIST'))


def reduce_metric_sums(sums, dist=None):
    """sums: float64 tensor [n_rows, n_metrics+1] = per-dataset (sum of seq_mean*n_seq per metric ..., sum n_seq).
    Returns the all-reduced numpy array (identity without a process group)."""
    if dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or force_collectives()):
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    return sums.cpu().numpy()


def assign_sequences(weights, world_size):
    """Longest-processing-time-first assignment of sequences to ranks.
    weights[i] = cost of sequence i (windows x padded pixels).  Returns list of index lists per rank;
    deterministic (ties broken by index) so every rank computes the same plan without communication."""
    order = sorted(range(len(weights)), key=lambda i: (-weights[i], i))
    loads = [0.0] * world_size
    plan = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        plan[r].append(i)
        loads[r] += weights[i]
    for p in plan:
        p.sort()
    return plan
