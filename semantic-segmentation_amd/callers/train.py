"""Data-parallel training step: `detseg/apis/train.py:64-105` (optimizer with per-module learning-rate groups),
the 'poly' learning-rate policy of the configs (`configs/nyuv2/danet_r50_depthdeform.py:77`, mmcv's
PolyLrUpdaterHook) and the gradient exchange of `detseg/core/utils/dist_utils.py` (SUM, then / world size).

One process per GPU; the batch is the only sharded axis.  Gradients are all-reduced in buckets
after backward by `dist_utils.allreduce_grads` (RCCL over xGMI under backend "nccl", gloo in the CPU tests).
"""
import torch
import torch.distributed as dist

from .. import dist_utils
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


def train_step(model, optimizer, batch, it=0, max_iters=1, ignore_index=255, bucket_size_mb=32, reducer=None,
               amp_dtype=None):
    """One iteration: poly lr, forward, loss, backward, gradient all-reduce when distributed, SGD update.
    `batch` = dict(img, HHA, gt_label) already on the device.  With `reducer` (an OverlappedGradReducer built on
    the model) the all-reduce runs under backward; otherwise after it, as in the reference.  Returns the loss.
    `amp_dtype` (torch.bfloat16): forward and loss under torch.autocast -- fp32 parameters and optimizer, 16-bit
    activations; the DDCN layers then run their native 16-bit forward and backward.  Not in the reference (fp32 only)."""
    poly_lr(optimizer, it, max_iters)
    optimizer.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=amp_dtype, enabled=amp_dtype is not None):
        losses = model(batch["img"], batch["gt_label"], batch.get("HHA"), ignore_index)
        loss = sum(v.float() for k, v in losses.items() if "loss" in k)
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
    last gradient of a bucket arrives the bucket becomes ready, and ready buckets are flattened and all-reduced
    asynchronously (RCCL runs them on its own stream, over xGMI, under the rest of backward) STRICTLY IN BUCKET
    ORDER: bucket i is issued only after buckets 0..i-1, and `finish()` issues whatever is left, also in order.
    Every rank therefore issues the same collectives in the same order whatever order its hooks fire in -- also when
    a rank leaves some parameter without a gradient (it is reduced as zeros).  `finish()` waits, divides by the
    world size and copies the results back into the .grad tensors; call it before optimizer.step().

    One backward per finish().  For gradient accumulation run the earlier backward passes inside `no_sync()`
    (hooks are ignored, gradients only accumulate) and the last one outside; a hook that fires for a bucket that
    was already issued raises instead of silently reducing a stale gradient.
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
        self._sync = True
        self._reset()
        self.issued_order = []      # bucket indices in the order their all-reduce was issued (last iteration; tests)
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _reset(self):
        self._pending = [len(b) for b in self.buckets]
        self._next = 0              # buckets below this index have been issued
        self._inflight = []         # (flat, work) of bucket i at position i

    def _launch_next(self):
        bucket = self.buckets[self._next]
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        dist_utils.drain_for_host_backend(flat, self.group)
        work = dist.all_reduce(flat, group=self.group, async_op=True)
        if flat.is_cuda and dist.get_backend(self.group) == "gloo":
            # the one-GPU rehearsal: no kernel of this process may be queued while gloo moves the bucket through the host
            # (dist_utils.drain_for_host_backend), so backward does not run on under it -- nothing overlaps here
            work.wait()
        self._inflight.append((flat, work))
        self.issued_order.append(self._next)
        self._next += 1

    def _on_grad(self, p):
        if not self._sync:
            return
        i = self._bucket_of[id(p)]
        if i < self._next or self._pending[i] <= 0:
            raise RuntimeError(
                "OverlappedGradReducer: a gradient arrived for a bucket that was already reduced in this iteration "
                "(a second backward before finish()); wrap all but the last backward in reducer.no_sync()")
        if self._next == 0 and not self._inflight:
            self.issued_order = []
        self._pending[i] -= 1
        while self._next < len(self.buckets) and self._pending[self._next] == 0:
            self._launch_next()

    def no_sync(self):
        """Context manager: backward passes inside only accumulate into .grad (no all-reduce is issued)."""
        reducer = self

        class _NoSync:
            def __enter__(self):
                reducer._sync = False

            def __exit__(self, *exc):
                reducer._sync = True
                return False
        return _NoSync()

    def finish(self):
        """Issue the buckets that are still missing (parameters without a gradient on this rank) in order, wait for
        every bucket, divide by the world size and write the means back."""
        if self._next == 0 and not self._inflight:
            self.issued_order = []
        while self._next < len(self.buckets):
            self._launch_next()
        for i, (flat, work) in enumerate(self._inflight):
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
        self._reset()

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []
