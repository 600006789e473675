"""Data-parallel gradient exchange for the DDCN layers: the only collective on the path.

Same contract as the reference's `allreduce_grads` / `_allreduce_coalesced`
(`detseg/core/utils/dist_utils.py:9-41`): every existing gradient is summed over the ranks and divided by the world
size, in place.  `torch.distributed` backend "nccl" is RCCL over xGMI on MI355X; the same code runs on gloo for the
CPU tests.  The reference's default packs every gradient of one tensor type into a single flat buffer, reduced
after backward; the default here is 32 MiB buckets, so that several RCCL calls pipeline over the 7 xGMI links
(`callers/train.py:OverlappedGradReducer` additionally issues them from inside backward).
"""
import torch
import torch.distributed as dist


def _split_into_buckets(tensors, bucket_size_mb):
    """Lists of same-dtype tensors: at most `bucket_size_mb` each, or one list per dtype when it is <= 0."""
    by_kind = {}
    for t in tensors:
        by_kind.setdefault((t.dtype, t.device), []).append(t)
    limit = bucket_size_mb * 1024 * 1024 if bucket_size_mb > 0 else None
    for group in by_kind.values():
        if limit is None:
            yield group
            continue
        bucket, used = [], 0
        for t in group:
            nbytes = t.numel() * t.element_size()
            if bucket and used + nbytes > limit:
                yield bucket
                bucket, used = [], 0
            bucket.append(t)
            used += nbytes
        if bucket:
            yield bucket


def drain_for_host_backend(tensor, group=None):
    """gloo moves a GPU tensor through host memory from a worker thread.  With the producing kernels still queued on the
    stream, two ranks that share ONE GPU (the CPU-backend rehearsal of the N-rank path) wait for each other forever
    (tools/gloo_probe.py); draining the stream first avoids it.  A no-op for RCCL ("nccl"), which is stream-ordered."""
    if tensor.is_cuda and dist.get_backend(group) == "gloo":
        torch.cuda.current_stream(tensor.device).synchronize()


def _allreduce_coalesced(tensors, world_size, bucket_size_mb=-1):
    for bucket in _split_into_buckets(tensors, bucket_size_mb):
        flat = torch.cat([t.reshape(-1) for t in bucket])
        drain_for_host_backend(flat)
        dist.all_reduce(flat)          # SUM ...
        flat.div_(world_size)          # ... then the mean, as the reference does
        start = 0
        for t in bucket:
            t.copy_(flat[start:start + t.numel()].view_as(t))
            start += t.numel()


def allreduce_grads(model_or_params, coalesce=True, bucket_size_mb=32):
    """All-reduce (mean) the .grad of every parameter that has one; `coalesce=False` reduces tensor by tensor."""
    params = model_or_params.parameters() if hasattr(model_or_params, "parameters") else model_or_params
    grads = [p.grad.data for p in params if p.requires_grad and p.grad is not None]
    world_size = dist.get_world_size()
    if coalesce:
        _allreduce_coalesced(grads, world_size, bucket_size_mb)
        return
    for g in grads:
        g.div_(world_size)
        drain_for_host_backend(g)
        dist.all_reduce(g)


def shard_batch(n_images, rank, world_size):
    """Contiguous partition of a global batch over ranks (images are the only shardable axis of the op)."""
    base, rem = divmod(n_images, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)
