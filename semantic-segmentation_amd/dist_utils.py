"""Data-parallel gradient exchange for the DDCN layers: the only collective on the path.

Mirrors `detseg/core/utils/dist_utils.py:9-41` of the reference (`allreduce_grads` /
`_allreduce_coalesced`): gradients are flattened into buckets per tensor type, all-reduced (SUM) and divided
by the world size, then copied back.  `torch.distributed` backend "nccl" is RCCL over xGMI on MI355X; the
same code runs on gloo for the CPU tests.  Unlike the reference's default (one bucket holding every
gradient, issued after backward), the default bucket here is 32 MiB so that several RCCL calls pipeline
over the 7 xGMI links.
"""
from collections import OrderedDict

import torch
import torch.distributed as dist
from torch._utils import _flatten_dense_tensors, _take_tensors, _unflatten_dense_tensors


def _allreduce_coalesced(tensors, world_size, bucket_size_mb=-1):
    # dist_utils.py:9-28
    if bucket_size_mb > 0:
        bucket_size_bytes = bucket_size_mb * 1024 * 1024
        buckets = _take_tensors(tensors, bucket_size_bytes)
    else:
        buckets = OrderedDict()
        for tensor in tensors:
            tp = tensor.type()
            if tp not in buckets:
                buckets[tp] = []
            buckets[tp].append(tensor)
        buckets = buckets.values()

    for bucket in buckets:
        flat_tensors = _flatten_dense_tensors(bucket)
        dist.all_reduce(flat_tensors)
        flat_tensors.div_(world_size)
        for tensor, synced in zip(bucket, _unflatten_dense_tensors(flat_tensors, bucket)):
            tensor.copy_(synced)


def allreduce_grads(model_or_params, coalesce=True, bucket_size_mb=32):
    """SUM-then-divide all-reduce of every existing gradient (dist_utils.py:31-41)."""
    params = model_or_params.parameters() if hasattr(model_or_params, "parameters") else model_or_params
    grads = [p.grad.data for p in params if p.requires_grad and p.grad is not None]
    world_size = dist.get_world_size()
    if coalesce:
        _allreduce_coalesced(grads, world_size, bucket_size_mb)
    else:
        for tensor in grads:
            dist.all_reduce(tensor.div_(world_size))


def shard_batch(n_images, rank, world_size):
    """Contiguous partition of a global batch over ranks (images are the only shardable axis of the op)."""
    base, rem = divmod(n_images, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)
