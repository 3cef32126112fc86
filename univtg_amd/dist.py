"""Data-parallel helpers (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm,
"gloo" in the CPU tests).  The hot path shards by sample with a single exchange step: the gradient all-reduce."""
from __future__ import annotations

import torch
import torch.distributed as dist


def allreduce_flat_(flat: torch.Tensor, bucket_elems: int, group=None, async_op: bool = False):
    """In-place SUM all-reduce of a flat buffer in a few large buckets (xGMI rings are per-link bound: prefer few,
    large collectives).  Returns the list of work handles when async_op."""
    n = flat.numel()
    works = []
    for o in range(0, n, max(1, bucket_elems)):
        w = dist.all_reduce(flat[o: min(n, o + bucket_elems)], op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            works.append(w)
    return works


def allreduce_many_(tensors, bucket_elems: int, group=None):
    """In-place SUM all-reduce of SEVERAL flat ranges as ONE collective where the backend can coalesce (RCCL: one group call, one kernel launch,
    one pair of stream hand-overs instead of one per range -- each costs 10-20 us of GPU-side latency on the communication stream); plain calls
    per range elsewhere (gloo in the CPU tests)."""
    tensors = [t for t in tensors if t.numel()]
    if not tensors:
        return
    backend = dist.get_backend(group)
    if len(tensors) > 1 and backend == "nccl":
        with dist._coalescing_manager(group=group, device=tensors[0].device, async_ops=False):
            for t in tensors:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return
    for t in tensors:
        allreduce_flat_(t, bucket_elems, group)


def broadcast_flat_(flat: torch.Tensor, src: int = 0, group=None):
    """DDP's initial parameter broadcast (main/train_vlp_ddp.py:272-275) on the flat parameter buffer."""
    dist.broadcast(flat, src=src, group=group)


def shard_batch(n_samples: int, rank: int, world: int):
    """DistributedSampler-style contiguous-stride sharding of sample indices (main/train_vlp_ddp.py:112)."""
    return list(range(rank, n_samples, world))
