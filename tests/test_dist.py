"""CPU, gloo, world_size 2: the data-parallel gradient exchange used by bench.py --gpus N and the batch
partition.  (On the GPU box the same code runs on RCCL; the op itself needs no collective.)"""
import importlib
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, coalesce, bucket_mb, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    du = importlib.import_module("semantic-segmentation_amd.dist_utils")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.Conv2d(4, 2, 1))   # same init on all ranks
    g = torch.Generator().manual_seed(100 + rank)
    for p in model.parameters():
        p.grad = torch.randn(p.shape, generator=g)
    local = [p.grad.clone() for p in model.parameters()]
    du.allreduce_grads(model, coalesce=coalesce, bucket_size_mb=bucket_mb)
    # expected: mean over ranks of the per-rank gradients (SUM then / world, dist_utils.py:22-28)
    exp = []
    for i, p in enumerate(model.parameters()):
        acc = torch.zeros_like(p)
        for r in range(world):
            gr = torch.Generator().manual_seed(100 + r)
            vals = [torch.randn(q_.shape, generator=gr) for q_ in model.parameters()]
            acc += vals[i]
        exp.append(acc / world)
    ok = all(torch.allclose(p.grad, e, atol=1e-6) for p, e in zip(model.parameters(), exp))
    changed = any(not torch.equal(p.grad, l) for p, l in zip(model.parameters(), local))
    q.put((rank, ok and changed, du.shard_batch(5, rank, world)))
    dist.destroy_process_group()


@pytest.mark.parametrize("coalesce,bucket_mb", [(True, -1), (True, 32), (False, -1)])
def test_allreduce_grads_gloo_world2(coalesce, bucket_mb):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, coalesce, bucket_mb, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert [s for _, _, s in res] == [(0, 3), (3, 5)]     # 5 images over 2 ranks, contiguous, no overlap


def test_shard_batch_covers_everything():
    du = importlib.import_module("semantic-segmentation_amd.dist_utils")
    for n in (1, 2, 7, 8, 16):
        for w in (1, 2, 4, 8):
            spans = [du.shard_batch(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
