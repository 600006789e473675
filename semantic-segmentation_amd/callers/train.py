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


def train_step(model, optimizer, batch, it=0, max_iters=1, ignore_index=255, bucket_size_mb=32):
    """One iteration: poly lr, forward, loss, backward, gradient all-reduce when distributed, SGD update.
    `batch` = dict(img, HHA, gt_label) already on the device.  Returns the (detached) loss."""
    poly_lr(optimizer, it, max_iters)
    optimizer.zero_grad(set_to_none=True)
    losses = model(batch["img"], batch["gt_label"], batch.get("HHA"), ignore_index)
    loss = sum(v for k, v in losses.items() if "loss" in k)
    loss.backward()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        allreduce_grads(model, coalesce=True, bucket_size_mb=bucket_size_mb)
    optimizer.step()
    return loss.detach()
