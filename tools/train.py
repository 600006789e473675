#!/usr/bin/env python3
"""Train a DDCN segmentor on synthetic data: the counterpart of the reference's `tools/train.py:15-109` (argument
names kept: config, --work_dir, --resume_from, --validate, --gpus, --seed, --launcher, --local_rank, --autoscale-lr)
and of `tools/dist_train.sh:8-9`.

    python tools/train.py danet_r50_depthdeform --gpus 1 --iters 20
    python tools/train.py danet_r101_depthdeform --gpus 8 --launcher pytorch        # starts its 8 ranks itself

`config` names one of the built-in configurations below (the reference's mmcv python configs cannot be loaded: mmcv is
absent); data are synthetic tensors of the configuration's shape (no dataset is reachable).  With `--launcher pytorch`
and no launcher environment the script starts `--gpus` ranks itself (semantic-segmentation_amd/launch.py), one process
per GPU over RCCL, BEFORE touching the GPU; under torchrun it is one rank of that job.  Checkpoints
(`--work_dir`, `--resume_from`) hold model + optimizer state and the iteration, as the reference's runner saves them.
"""
import argparse
import importlib
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {
    # configs/nyuv2/danet_r50_depthdeform.py:3-23,33,47,60-77
    "danet_r50_depthdeform": dict(depth=50, deep_base=False, classes=40, crop=(480, 640), imgs_per_gpu=4, lr=0.00025,
                                  norm="BN"),
    # configs/sunrgbd/danet_r101_depthdeform.py:3-27,34-37,51 (SyncBN)
    "danet_r101_depthdeform": dict(depth=101, deep_base=True, classes=37, crop=(500, 500), imgs_per_gpu=2, lr=0.00025,
                                   norm="SyncBN"),
}


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Train a segmentor")
    parser.add_argument("config", choices=sorted(CONFIGS), help="built-in train configuration")
    parser.add_argument("--work_dir", help="the dir to save logs and models")
    parser.add_argument("--resume_from", help="the checkpoint file to resume from")
    parser.add_argument("--validate", action="store_true", help="accepted for compatibility; no dataset to validate on")
    parser.add_argument("--gpus", type=int, default=1, help="number of gpus (ranks) to use")
    parser.add_argument("--seed", type=int, default=None, help="random seed")
    parser.add_argument("--launcher", choices=["none", "pytorch", "slurm", "mpi"], default="none", help="job launcher")
    parser.add_argument("--local_rank", type=int, default=0)
    parser.add_argument("--autoscale-lr", action="store_true", help="scale lr with the number of gpus (lr * gpus / 8)")
    parser.add_argument("--iters", type=int, default=20, help="training iterations of THIS run (the reference counts epochs)")
    parser.add_argument("--max_iters", type=int, default=None,
                        help="length of the whole poly schedule (the reference's total_epochs): fixed across --resume_from, "
                             "stored in the checkpoint; default: --iters of the first run")
    parser.add_argument("--crop", type=int, nargs=2, default=None, metavar=("H", "W"), help="override the crop size")
    parser.add_argument("--imgs_per_gpu", type=int, default=None)
    parser.add_argument("--log_interval", type=int, default=10)
    args = parser.parse_args(argv)
    if "LOCAL_RANK" not in os.environ:
        os.environ["LOCAL_RANK"] = str(args.local_rank)
    return args


def _launch_module():
    spec = importlib.util.spec_from_file_location("ddcn_launch", os.path.join(ROOT, "semantic-segmentation_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main(argv=None):
    args = parse_args(argv)
    launch = _launch_module()
    if args.launcher in ("slurm", "mpi"):
        raise NotImplementedError("launcher %r: only 'none' and 'pytorch' are built" % args.launcher)
    if args.launcher == "pytorch" and not launch.under_launcher():
        import torch
        if torch.cuda.device_count() < args.gpus:       # counting devices does not initialise the GPU
            raise SystemExit("train.py --gpus %d: only %d GPU(s) visible" % (args.gpus, torch.cuda.device_count()))
        sys.exit(launch.spawn_ranks(args.gpus, sys.argv[1:] if argv is None else list(argv),
                                    script=os.path.abspath(__file__)))

    import torch
    import torch.distributed as dist
    callers = importlib.import_module("semantic-segmentation_amd.callers")
    cfg = dict(CONFIGS[args.config])
    distributed = args.launcher == "pytorch"
    if distributed:
        rank, world, dev = launch.init_dist("pytorch", backend="nccl")       # detseg/apis/env.py:28-33
    else:
        rank, world, dev = 0, 1, torch.device("cuda", 0)
        torch.cuda.set_device(dev)
    if args.seed is not None:                                                 # detseg/apis/env.py:55-59
        import random
        import numpy as np
        random.seed(args.seed); np.random.seed(args.seed); torch.manual_seed(args.seed); torch.cuda.manual_seed_all(args.seed)
    else:
        torch.manual_seed(0)                                                  # same initial weights on every rank
    lr = cfg["lr"] * (args.gpus / 8 if args.autoscale_lr else 1.0)
    h, w = args.crop or cfg["crop"]
    n = args.imgs_per_gpu or cfg["imgs_per_gpu"]

    backbone = callers.ResNet(cfg["depth"], strides=(1, 2, 1, 1), dilations=(1, 1, 2, 4),
                              dcn=dict(dcn_type="depthdeform"), stage_with_dcn=(False, True, True, True),
                              norm_eval=False, deep_base=cfg["deep_base"])
    model = callers.DANet(backbone, callers.DANetHead(2048, cfg["classes"]), callers.PointNet()).to(dev).train()
    if distributed and world > 1 and cfg["norm"] == "SyncBN":
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    opt = callers.build_optimizer(model, lr=lr)
    start, max_iters = 0, args.max_iters
    if args.resume_from:
        ck = torch.load(args.resume_from, map_location=dev, weights_only=True)
        model.load_state_dict(ck["state_dict"])
        opt.load_state_dict(ck["optimizer"])
        start = int(ck["meta"]["iter"])
        # the schedule's length is the first run's (mmcv's runner keeps total_epochs across resume): the learning rate
        # continues on the SAME polynomial instead of restarting a new one over start + iters
        if max_iters is None and ck["meta"].get("max_iters"):
            max_iters = int(ck["meta"]["max_iters"])
    reducer = callers.OverlappedGradReducer(model) if world > 1 else None
    g = torch.Generator().manual_seed(1234 + rank)

    def batch():
        label = torch.randint(0, cfg["classes"], (n, h, w), generator=g)
        label[torch.rand(n, h, w, generator=g) < 0.05] = 255
        return dict(img=torch.randn(n, 3, h, w, generator=g).to(dev), HHA=torch.randn(n, 3, h, w, generator=g).to(dev),
                    gt_label=label.to(dev))

    total = start + args.iters
    if max_iters is None:
        max_iters = total
    total = min(total, max_iters)                       # the schedule ends at max_iters (lr reaches 0 there)
    t0 = time.perf_counter()
    loss = None
    for it in range(start, total):
        loss = callers.train_step(model, opt, batch(), it, max_iters, ignore_index=255, reducer=reducer)
        if rank == 0 and ((it + 1) % args.log_interval == 0 or it + 1 == total):
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(json.dumps({"iter": it + 1, "loss": round(float(loss), 4), "lr": opt.param_groups[0]["lr"],
                              "images_per_s": round(n * world * (it + 1 - start) / dt, 2), "world_size": world}),
                  flush=True)
    if args.work_dir and rank == 0:
        os.makedirs(args.work_dir, exist_ok=True)
        path = os.path.join(args.work_dir, "iter_%d.pth" % total)
        torch.save({"state_dict": model.state_dict(), "optimizer": opt.state_dict(),
                    "meta": {"iter": total, "max_iters": max_iters, "config": args.config}}, path)
        print(json.dumps({"checkpoint": path}), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
