"""Data-parallel training step: `detseg/apis/train.py:64-105` (optimizer with per-module learning-rate groups),
the 'poly' learning-rate policy of the configs (`configs/nyuv2/danet_r50_depthdeform.py:77`, mmcv's
PolyLrUpdaterHook) and the gradient exchange of `detseg/core/utils/dist_utils.py` (SUM, then / world size).

One process per GPU; the batch is the only sharded axis.  Gradients are all-reduced in buckets
after backward by `dist_utils.allreduce_grads` (RCCL over xGMI under backend "nccl", gloo in the CPU tests).
"""
import torch
import torch.distributed as dist

from ..dist_utils import allreduce_grads

# configs/*/danet_*_depthdeform.py: backbone x1, head x10, depth backbone x10
PARAMGROUPS = (dict(params="backbone", lr_mult=1), dict(params="head", lr_mult=10),
               dict(params="backbone_depth", lr_mult=10))


def build_optimizer(model, lr=0.00025, momentum=0.9, weight_decay=0.0001, paramgroup_options=PARAMGROUPS):
    """SGD with one parameter group per named sub-module, lr = base * lr_mult (train.py:96-104)."""
    model = getattr(model, "module", model)
    groups = []
    for opt in paramgroup_options:
        sub = getattr(model, opt["params"], None)
        if sub is None:
            continue
        groups.append(dict(params=list(sub.parameters()), lr=lr * opt["lr_mult"], initial_lr=lr * opt["lr_mult"]))
    return torch.optim.SGD(groups, lr=lr, momentum=momentum, weight_decay=weight_decay)


def poly_lr(optimizer, it, max_iters, power=0.9, min_lr=0.0):
    """lr_i = (base_i - min_lr) * (1 - it / max_iters) ** power + min_lr for every group, by iteration."""
    coeff = (1 - it / max_iters) ** power
    for group in optimizer.param_groups:
        group["lr"] = (group["initial_lr"] - min_lr) * coeff + min_lr


def train_step(model, optimizer, batch, it=0, max_iters=1, ignore_index=255, bucket_size_mb=32, reducer=None):
    """One iteration: poly lr, forward, loss, backward, gradient all-reduce when distributed, SGD update.
    `batch` = dict(img, HHA, gt_label) already on the device.  With `reducer` (an OverlappedGradReducer built on
    the model) the all-reduce runs under backward; otherwise after it, as in the reference.  Returns the loss."""
    poly_lr(optimizer, it, max_iters)
    optimizer.zero_grad(set_to_none=True)
    losses = model(batch["img"], batch["gt_label"], batch.get("HHA"), ignore_index)
    loss = sum(v for k, v in losses.items() if "loss" in k)
    loss.backward()
    if reducer is not None:
        reducer.finish()
    elif dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        allreduce_grads(model, coalesce=True, bucket_size_mb=bucket_size_mb)
    optimizer.step()
    return loss.detach()


class OverlappedGradReducer:
    """Backward-overlapped, bucketed gradient all-reduce with the reference's SUM-then-divide semantics.

    The reference reduces after backward has finished (`_allreduce_coalesced`, dist_utils.py:9-28, called from its
    optimizer hook).  Here every parameter gets a post-accumulate-grad hook; parameters are assigned to buckets of
    `bucket_size_mb` in REVERSE registration order (roughly the order backward produces their gradients); when the
    last gradient of a bucket arrives the bucket is flattened and its all-reduce is issued asynchronously (RCCL
    runs it on its own stream, over xGMI, under the rest of backward).  `finish()` waits, divides by the world
    size and copies the results back into the .grad tensors; call it before optimizer.step().  A parameter that
    receives no gradient in an iteration is reduced as zeros, so every rank issues the same collectives.
    """

    def __init__(self, model_or_params, bucket_size_mb=32, process_group=None):
        params = model_or_params.parameters() if hasattr(model_or_params, "parameters") else model_or_params
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        limit = int(bucket_size_mb * 1024 * 1024)
        self.buckets, cur, size = [], [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * p.element_size()
            if cur and (size + nbytes > limit or p.dtype != cur[0].dtype):
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        self._pending = [len(b) for b in self.buckets]
        self._inflight = {}
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _launch(self, i):
        bucket = self.buckets[i]
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        self._inflight[i] = (flat, dist.all_reduce(flat, group=self.group, async_op=True))

    def _on_grad(self, p):
        i = self._bucket_of[id(p)]
        self._pending[i] -= 1
        if self._pending[i] == 0:
            self._launch(i)

    def finish(self):
        """Wait for every bucket, divide by the world size, write back; buckets whose hooks did not all fire
        (unused parameters) are reduced now."""
        for i in range(len(self.buckets)):
            if i not in self._inflight:
                self._launch(i)
        for i, (flat, work) in sorted(self._inflight.items()):
            work.wait()
            flat.div_(self.world)
            o = 0
            for p in self.buckets[i]:
                n = p.numel()
                if p.grad is None:
                    p.grad = flat[o:o + n].view_as(p).clone()
                else:
                    p.grad.copy_(flat[o:o + n].view_as(p))
                o += n
        self._inflight.clear()
        self._pending = [len(b) for b in self.buckets]

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []
