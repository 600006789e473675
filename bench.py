#!/usr/bin/env python3
"""bench.py -- DDCN operator forward+backward on MI355X (BASELINE.json metric "DDCN op fwd+bwd GB/s & %MFMA").

A step = one forward + one backward of the modulated DDCN op (the main op of the reference's
DepthDeformConvPack, detseg/ops/dcn/deform_conv.py:457-463) through the autograd.Function boundary, on
synthetic inputs already resident in HBM.  Workload = BASELINE.json configs[0] layer: ResNet res5 shape,
N=2 per GPU, C=512->512, 97x97, 3x3, stride 1, pad=dilation=4, groups=1, deformable_groups=1, fp32.
`--workload r50` / `--workload r101` run the whole models of BASELINE configs[1] / configs[2] instead (DDCN-ResNet50
NYUv2 480x640 batch 4; DDCN-ResNet101 769^2 batch 8 -- SURVEY.md 8(f3)/(f2) callers): a step is then forward +
backward + gradient all-reduce (N > 1) + SGD update, and the line reports train images/s.  The default stays the
layer: it is the workload the op metric (GB/s, %MFMA) and the per-kernel rooflines are defined on.

value = algorithmic gather/scatter bytes of the reference algorithm per step (SURVEY.md 8(d):
775.8 MB/image = 2*B_im2col + B_col2im + B_coord) * images / time.  The op is MFMA-bound overall, so the
JSON line also carries op TFLOP/s and a roofline entry per kernel (measured live with HIP events on the
launch stream).  With --gpus N every rank runs the same per-GPU batch (weak scaling, images partitioned over
ranks) and all-reduces grad_weight (SUM, then /world) as the reference's data-parallel training does
(detseg/core/utils/dist_utils.py:22-28).

Launch: `python bench.py --gpus N` starts its N ranks itself (one process per GPU, RCCL) BEFORE anything touches the
GPU in the parent (semantic-segmentation_amd/launch.py, the counterpart of tools/dist_train.sh:8-9); under torchrun
(WORLD_SIZE set) it is one rank of that job.  `--workload r101 --gpus N [--syncbn]` is BASELINE configs[3].
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _load_launch():
    """semantic-segmentation_amd/launch.py by path: the parent of an N-rank job must not import the package (that
    would load the HIP library) before it has started its ranks."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ddcn_launch", os.path.join(ROOT, "semantic-segmentation_amd",
                                                                              "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3   # MI355X_MICROARCH.md: FP32 matrix 157.3 TFLOP/s


def workload(args):
    return dict(N=args.batch, C=args.channels, Co=args.channels, H=args.size, W=args.size, k=3, stride=1,
                pad=args.dilation, dil=args.dilation, groups=1, dg=1)


def alg_numbers(w):
    """SURVEY.md 8(d) algorithmic bytes / flops per image (fp32, modulated)."""
    e, C, K2 = 4, w["C"], w["k"] * w["k"]
    P = w["H"] * w["W"]  # stride 1, pad = dil -> Ho = H
    HW = w["H"] * w["W"]
    dg = w["dg"]
    b_im2col = e * (C * HW + 3 * dg * K2 * P + C * K2 * P)
    b_col2im = b_im2col
    b_coord = e * (C * K2 * P + C * HW + 3 * dg * K2 * P + 3 * dg * K2 * P)
    b_fused = e * (C * K2 * P + 2 * C * HW + 2 * 3 * dg * K2 * P)
    f_gemm = 2.0 * w["Co"] * C * K2 * P
    return dict(b_im2col=b_im2col, b_col2im=b_col2im, b_coord=b_coord, b_fused=b_fused, f_gemm=f_gemm,
                b_total=2 * b_im2col + b_col2im + b_coord, f_total=3 * f_gemm)


def make_inputs(w, dev, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    N, C, H, W = w["N"], w["C"], w["H"], w["W"]
    K = C * 9
    x = torch.randn(N, C, H, W, generator=g)
    weight = (torch.rand(w["Co"], C, 3, 3, generator=g) * 2 - 1) / (K ** 0.5)
    offset = torch.randn(N, 18, H, W, generator=g) * w.get("offstd", 2)
    mask = torch.sigmoid(torch.randn(N, 9, H, W, generator=g))
    gout = torch.randn(N, w["Co"], H, W, generator=g)
    return [t.to(dev) for t in (x, weight, offset, mask, gout)]


LDS_ATOMIC64_CYCLES, N_CU, CLOCK_HZ = 19, 256, 2.4e9   # measured wave-instruction cost; MI355X_MICROARCH.md
PREHEAT_STEPS = 20      # untimed steps before the warm-up of the layer workload (see main)


def time_stage(fn, iters, warmup=2):
    for _ in range(warmup):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters  # ms per launch


def kernel_rooflines(w, tensors, iters):
    """Average launch duration of every stage kernel (HIP events on the launch stream = torch's current)."""
    _lib = importlib.import_module("semantic-segmentation_amd.dcn._lib")
    L = _lib.lib()
    x, weight, offset, mask, gout = tensors
    N = w["N"]
    g = _lib.Geom(N, w["C"], w["H"], w["W"], w["Co"], 3, 3, 1, 1, w["pad"], w["pad"], w["dil"], w["dil"], 1, 1,
                  w["H"], w["W"])
    col = torch.empty(L.ddcn_workspace_bytes(g) // 4, device=x.device)
    out = torch.empty_like(gout)
    gx, goff, gmask, gw = (torch.zeros_like(t) for t in (x, offset, mask, weight))
    s = _lib.current_stream(x.device)
    p = _lib.ptr
    a = alg_numbers(w)
    stages = [
        ("im2col", "hbm", a["b_im2col"] * N,
         lambda: _lib.check(L.ddcn_im2col_f32(p(x), p(offset), p(mask), p(col), g, s))),
        ("gemm_out", "mfma", a["f_gemm"] * N,
         lambda: _lib.check(L.ddcn_gemm_out_f32(p(weight), p(col), None, p(out), g, s))),
        ("gemm_gcol", "mfma", a["f_gemm"] * N,
         lambda: _lib.check(L.ddcn_gemm_gcol_f32(p(weight), p(gout), p(col), g, s))),
        ("col2im_coord", "hbm", a["b_fused"] * N,
         lambda: _lib.check(L.ddcn_col2im_coord_f32(p(col), p(x), p(offset), p(mask), p(gx), p(goff), p(gmask), g, s))),
        ("gemm_gweight", "mfma", a["f_gemm"] * N,
         lambda: _lib.check(L.ddcn_gemm_gweight_f32(p(gout), p(col), p(gw), g, s))),
    ]
    # HBM bytes per launch come from the rocprofv3 --pmc passes committed under profiles/ (counters cannot be read from
    # inside this process).  The file records the commit and a digest of csrc/ at collection time: when the kernels'
    # sources have changed since, the figure is stale and is dropped (traffic = null) instead of being reported.
    traffic, traffic_src = committed_traffic(w)
    res = []
    for name, bound, work, fn in stages:
        ms = time_stage(fn, iters)
        if bound == "hbm":
            ach, peak, unit = work / ms / 1e6, HBM_PEAK_GBS, "GB/s"
        else:
            ach, peak, unit = work / ms / 1e9, MFMA_F32_PEAK_TF, "TFLOP/s"
        res.append(dict(kernel=name, bound=bound, ms=round(ms, 4), achieved=round(ach, 2), peak=peak, unit=unit,
                        frac=round(ach / peak, 4), traffic=traffic.get(name), work=work))
        if name == "col2im_coord":
            res[-1]["lds_atomic"] = lds_atomic_floor(w, ms)
    return res, traffic_src


def lds_atomic_floor(w, ms):
    """The scatter's real bound is the LDS-atomic rate, not HBM: four atomics per (pixel, tap, channel) into
    the workgroup's LDS footprint, two channels per ds_add_u64 in the default packed fixed-point format
    (DDCN_SCATTER_FX=0: one ds_add_u64 per channel); a CU retires one ds_add_u64 wave instruction per ~19 cycles whatever
    the addresses (tools/ubench_lds_atomic.hip).  -> the floor in ms and the measured launch as a fraction of it."""
    packed = os.environ.get("DDCN_SCATTER_FX", "1") != "0"
    winstr = w["N"] * w["C"] * w["H"] * w["W"] * 9 * 4 / 64 / (2 if packed else 1)
    floor_ms = winstr * LDS_ATOMIC64_CYCLES / (N_CU * CLOCK_HZ) * 1e3
    return {"wave_instructions": int(winstr), "cycles_per_instruction_per_cu": LDS_ATOMIC64_CYCLES,
            "floor_ms": round(floor_ms, 4), "frac": round(floor_ms / ms, 4),
            "format": "2 channels per ds_add_u64" if packed else "1 channel per ds_add_u64"}


def csrc_digest():
    """sha256 over the kernel sources (csrc/*.hip, *.h and the public header), in name order."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "semantic-segmentation_amd", "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")))
    files.append(os.path.join(ROOT, "include", "ddcn_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def committed_traffic(w):
    """({kernel: HBM bytes per launch}, provenance dict) from the newest profiles/*_hbm_traffic.json that belongs to
    this workload AND to the kernel sources in the tree; ({}, reason) otherwise."""
    import glob
    if (w["N"], w["C"], w["Co"], w["H"], w["W"], w["dil"]) != (2, 512, 512, 97, 97, 4):
        return {}, {"file": None, "why": "PMC passes are recorded for the default workload only"}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
    if not files:
        return {}, {"file": None, "why": "no committed PMC file"}
    with open(files[-1]) as fh:
        rec = json.load(fh)
    src = {"file": os.path.relpath(files[-1], ROOT), "commit": rec.get("commit"), "csrc_sha256": rec.get("csrc_sha256")}
    if rec.get("csrc_sha256") != csrc_digest():
        src["why"] = "stale: csrc/ changed since the counters were collected; figure dropped"
        return {}, src
    return {k: v["traffic_bytes"] for k, v in rec["kernels"].items()}, src


def host_cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(w):
    """The reference algorithm on the host cores, whole workload (all N images), fwd+bwd, fp32: the C
    oracle's OpenMP gather / scatter stages (oracle/ddcn_ref_impl.inc, a port of the CUDA kernels; the reference has
    no CPU path) and torch.mm (the host BLAS) for G1-G3, composed per image in the reference's order
    (deform_conv_cuda.cpp:419-448 forward, :497-561 backward).  Threads = the process's CPU share: OMP_NUM_THREADS when
    the environment sets one (16 on the 1-GPU box, whose os.cpu_count() shows the whole machine), else every core."""
    from oracle import ddcn_oracle as orc
    import numpy as np
    host = os.cpu_count() or 1
    threads = max(1, min(host, int(os.environ.get("OMP_NUM_THREADS", host))))
    orc.set_threads(threads)
    torch.set_num_threads(threads)
    rng = np.random.default_rng(0)
    N, C, Co, H, W = w["N"], w["C"], w["Co"], w["H"], w["W"]
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    weight = (rng.uniform(-1, 1, (Co, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)
    offset = (rng.standard_normal((N, 18, H, W)) * w.get("offstd", 2)).astype(np.float32)
    mask = (1 / (1 + np.exp(-rng.standard_normal((N, 9, H, W))))).astype(np.float32)
    gout = rng.standard_normal((N, Co, H, W)).astype(np.float32)
    geo = (3, 3, 1, w["pad"], w["dil"], w["dg"])
    wm = torch.from_numpy(weight.reshape(Co, C * 9))
    P = H * W

    def step():
        out = np.empty((N, Co, P), dtype=np.float32)
        gx = np.empty_like(x)
        goff = np.empty_like(offset)
        gmask = np.empty_like(mask)
        gw = torch.zeros(Co, C * 9)
        for n in range(N):                                   # forward, per image as the reference loops
            col = orc.im2col(x[n], offset[n], mask[n], *geo)
            torch.mm(wm, torch.from_numpy(col), out=torch.from_numpy(out[n]))
        for n in range(N):                                   # backward
            go = torch.from_numpy(gout[n].reshape(Co, P))
            gcol = torch.mm(wm.t(), go).numpy()                                       # G2
            goff[n], gmask[n] = orc.col2im_coord(gcol, x[n], offset[n], mask[n], *geo)
            gx[n] = orc.col2im(gcol, offset[n], mask[n], C, H, W, *geo)
            col = orc.im2col(x[n], offset[n], mask[n], *geo)                          # recomputed, as the reference does
            gw.addmm_(go, torch.from_numpy(col).t())                                  # G3
        return out

    step()                       # warm-up: page faults, BLAS thread pools
    times = []
    for _ in range(2):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    dt = min(times)
    a = alg_numbers(w)
    return dict(value=round(a["b_total"] * N / dt / 1e9, 4), unit="GB/s", cores=threads, kind="port",
                sample="the whole workload (N=%d images of the layer), fwd+bwd, fp32: C oracle gather/scatter (OpenMP) + "
                       "torch.mm for G1-G3; best of 2 passes after a warm-up, %.2f s per pass; host: %s, os.cpu_count()=%d, "
                       "threads used=%d" % (N, dt, host_cpu_model(), host, threads),
                tflops=round(a["f_total"] * N / dt / 1e12, 4))


MODEL_WORKLOADS = {
    # BASELINE configs[1]: configs/nyuv2/danet_r50_depthdeform.py (4 images per GPU, 40 classes, plain BN)
    "r50": dict(depth=50, deep_base=False, batch=4, h=480, w=640, classes=40,
                name="BASELINE configs[1]: DDCN-ResNet50 + DANet head, NYUv2 480x640 RGB + HHA"),
    # BASELINE configs[2]: configs/sunrgbd/danet_r101_depthdeform.py with deep_base, 19 classes, 769^2 crops
    "r101": dict(depth=101, deep_base=True, batch=8, h=769, w=769, classes=19,
                 name="BASELINE configs[2]: DDCN-ResNet101 (deep stem) + DANet head, Cityscapes-style 769x769 crop"),
}


def model_workload(args, world, rank, dev):
    """Training step of a whole DDCN model: forward, loss, backward, bucketed gradient
    all-reduce over RCCL when world > 1, SGD(momentum, wd) update with the poly schedule."""
    callers = importlib.import_module("semantic-segmentation_amd.callers")
    cfg = MODEL_WORKLOADS[args.workload]
    cfg = dict(cfg)
    n = args.batch if args.batch != 2 else cfg["batch"]     # --batch overrides the config's images per GPU
    if args.hw:
        cfg["h"], cfg["w"] = args.hw
        cfg["name"] += " [input size overridden to %dx%d: a rehearsal, not the BASELINE workload]" % tuple(args.hw)
    torch.manual_seed(0)                                    # same initial weights on every rank
    backbone = callers.ResNet(cfg["depth"], strides=(1, 2, 1, 1), dilations=(1, 1, 2, 4),
                              dcn=dict(dcn_type="depthdeform"), stage_with_dcn=(False, True, True, True),
                              norm_eval=False, deep_base=cfg["deep_base"])
    model = callers.DANet(backbone, callers.DANetHead(2048, cfg["classes"]), callers.PointNet(),
                          dcn_offset_init=args.offset_init).to(dev).train()
    if world > 1 and args.syncbn:
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    opt = callers.build_optimizer(model)
    # N > 1: gradient buckets are all-reduced (RCCL) from inside backward; --sync-allreduce reduces after it
    reducer = callers.OverlappedGradReducer(model) if (world > 1 and not args.sync_allreduce) else None
    g = torch.Generator().manual_seed(1234 + rank)
    label = torch.randint(0, cfg["classes"], (n, cfg["h"], cfg["w"]), generator=g)
    label[torch.rand(n, cfg["h"], cfg["w"], generator=g) < 0.05] = 255      # 5 % ignored (configs: ignore_index=255)
    batch = dict(img=torch.randn(n, 3, cfg["h"], cfg["w"], generator=g).to(dev),
                 HHA=torch.randn(n, 3, cfg["h"], cfg["w"], generator=g).to(dev), gt_label=label.to(dev))
    total = args.warmup + args.steps
    it = 0
    amp = torch.bfloat16 if args.amp == "bf16" else None
    for _ in range(args.warmup):
        callers.train_step(model, opt, batch, it, total, reducer=reducer, amp_dtype=amp)
        it += 1
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = callers.train_step(model, opt, batch, it, total, reducer=reducer, amp_dtype=amp)
        it += 1
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = max_over_ranks(dt, dev)
    devices = job_devices(dev, world)
    grad_bytes = sum(p.numel() * p.element_size() for p in model.parameters() if p.requires_grad)
    probe = allreduce_probe(dev, world, grad_bytes, iters=5) if world > 1 else None
    # every bucket size of the overlapped reducer, alone: the per-message times and bus bandwidths behind the step time
    # (one ring over xGMI is per-link bound, ~153 GB/s; all 7 links of a GPU give ~1 TB/s aggregate)
    bucket_probe = None
    if world > 1:
        tmp = reducer or callers.OverlappedGradReducer(model)
        sizes = [sum(p.numel() * p.element_size() for p in b) for b in tmp.buckets]
        if reducer is None:
            tmp.remove()
        bucket_probe = []
        for nbytes in sorted(set(sizes), reverse=True):
            r = allreduce_probe(dev, world, nbytes, iters=3)
            r["count"] = sizes.count(nbytes)
            bucket_probe.append(r)
    if rank == 0:
        print(json.dumps({
            "metric": "train images/sec", "value": round(n * world * args.steps / dt, 3), "unit": "images/s",
            "n_gpus": dist.get_world_size() if world > 1 else 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if amp is None else "bf16 activations under autocast, fp32 parameters",
            "data": "synthetic",
            "config": {"workload": cfg["name"] + ", %d images/GPU, fwd+bwd+SGD step, random init" % n,
                       "images_per_gpu": n, "parallelism": "dp%d" % world,
                       "params_M": round(sum(p.numel() for p in model.parameters()) / 1e6, 2),
                       "dcn_offset_init": args.offset_init,
                       "norm": "SyncBN" if (world > 1 and args.syncbn) else "BN",
                       "allreduce": "none" if world == 1 else ("after backward" if reducer is None else
                                                               "32 MiB buckets, overlapped with backward")},
            "final_loss": round(float(loss), 4),
            "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
            "devices": devices,
            "allreduce": None if probe is None else {
                "gradient": probe, "buckets": bucket_probe,
                "backend": "nccl (RCCL)" if args.backend == "nccl" else "gloo (rehearsal)",
                "distinct_devices": len({tuple(d[1:]) for d in devices})},
        }), flush=True)


def inference_workload(args, world, rank, dev):
    """BASELINE configs[4]: modulated DDCN (offset + mask, deformable_groups 4) in every ResNet stage, bf16
    activations / weights with fp32 accumulation, 1024x2048 inference, one image per step and GPU (replicas: the
    path does not shard below an image).  DDCN layers run the native bf16 kernels (ddcn_bf16.hip)."""
    callers = importlib.import_module("semantic-segmentation_amd.callers")
    torch.manual_seed(0)
    backbone = callers.ResNet(101, strides=(1, 2, 1, 1), dilations=(1, 1, 2, 4),
                              dcn=dict(dcn_type="modulated", deformable_groups=4),
                              stage_with_dcn=(True, True, True, True), norm_eval=True)
    model = callers.DANet(backbone, callers.DANetHead(2048, 19), None, dcn_offset_init=args.offset_init)
    model = model.to(dev).bfloat16().eval()
    n = 1 if args.batch == 2 else args.batch
    g = torch.Generator().manual_seed(1234 + rank)
    img = torch.randn(n, 3, 1024, 2048, generator=g).to(dev).bfloat16()
    graphed = False
    with torch.no_grad():
        for _ in range(max(args.warmup, 1)):
            out = model(img)
        torch.cuda.synchronize()
        run = lambda: model(img)
        if not args.no_graph:
            # ~900 short launches per image: replay them as one HIP graph (the C ABI neither allocates nor
            # synchronises, and the scratch tensors come from the allocator's capture pool)
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    gout = model(img)
                graph.replay()
                torch.cuda.synchronize()
                if not torch.equal(gout, out):
                    raise RuntimeError("graph replay differs from eager run")
                out, graphed = gout, True
                run = lambda: (graph.replay(), gout)[1]
            except Exception as e:   # keep the eager path measurable if capture is refused
                print("bench: HIP graph capture failed (%s); running eagerly" % e, file=sys.stderr)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = run()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    if world > 1:
        dt = max_over_ranks(dt, dev)
    if rank == 0:
        n_dcn = sum(1 for m in model.modules() if type(m).__name__ == "ModulatedDeformConvPack")
        print(json.dumps({
            "metric": "inference images/sec", "value": round(n * world * args.steps / dt, 3), "unit": "images/s",
            "n_gpus": dist.get_world_size() if world > 1 else 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4]: ResNet101 with modulated DDCN (dg=4) in all %d bottlenecks + "
                                   "DANet head, 1024x2048 inference, %d image/GPU, random init" % (n_dcn, n),
                       "images_per_gpu": n, "parallelism": "replicas%d" % world, "dcn_offset_init": args.offset_init,
                       "launch": "one HIP graph per image" if graphed else "eager"},
            "output_shape": list(out.shape), "finite": bool(torch.isfinite(out.float()).all()),
            "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
        }), flush=True)


def max_over_ranks(value, dev):
    """MAX of a python float over the job's ranks.  Under RCCL the scalar lives on the GPU (the backend has no CPU path);
    under gloo -- the rehearsal with several ranks on ONE GPU -- it stays on the host: a GPU tensor handed to gloo while
    this process still has work queued on the shared device is the deadlock dist_utils.drain_for_host_backend describes
    (it hung the rehearsal once the per-bucket probes made such calls frequent)."""
    on_gpu = dist.get_backend() != "gloo"
    t = torch.tensor([value], device=dev if on_gpu else "cpu", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_probe(dev, world, nbytes, iters=10):
    """One fp32 all-reduce (SUM) of `nbytes` over the job's ranks, alone: ms per call (max over ranks is implied by
    the collective) and bus bandwidth 2(N-1)/N * bytes / time, the figure comparable to the 7 x ~153 GB/s xGMI links."""
    drain = importlib.import_module("semantic-segmentation_amd.dist_utils").drain_for_host_backend   # no-op under RCCL
    buf = torch.ones(nbytes // 4, device=dev)
    for _ in range(2):
        drain(buf)
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        drain(buf)
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    dt = max_over_ranks(dt, dev)
    return dict(bytes=nbytes, ms=round(dt * 1e3, 4), bus_GBps=round(2 * (world - 1) / world * nbytes / dt / 1e9, 2))


def job_devices(dev, world):
    """What each rank really runs on: [rank, local device index, device name, uuid or PCI bus id], gathered on every rank."""
    props = torch.cuda.get_device_properties(dev)
    ident = str(getattr(props, "uuid", "")) or str(getattr(props, "pci_bus_id", ""))
    mine = [dist.get_rank() if world > 1 else 0, dev.index, props.name, ident]
    if world == 1:
        return [mine]
    out = [None] * world
    dist.all_gather_object(out, mine)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2, help="images per GPU")
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--size", type=int, default=97)
    ap.add_argument("--dilation", type=int, default=4)
    ap.add_argument("--workload", choices=("layer", "r50", "r101", "cfg5"), default="layer")
    ap.add_argument("--offset-init", choices=("zero", "kaiming"), default="zero",
                    help="model workloads: DDCN offset convolutions start at zero (the reference's intent) or Kaiming "
                         "(what its init_weights really does: ~6 px offsets)")
    ap.add_argument("--syncbn", action="store_true",
                    help="model workloads, N > 1: SyncBatchNorm as in configs/sunrgbd/danet_r101_depthdeform.py "
                         "(per-layer statistics exchanged over RCCL); default is plain BN per rank")
    ap.add_argument("--sync-allreduce", action="store_true",
                    help="model workloads, N > 1: all-reduce after backward (the reference's order) instead of under it")
    ap.add_argument("--no-graph", action="store_true", help="cfg5: launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="N > 1: 'nccl' = RCCL over xGMI, one GPU per rank (the measured configuration); 'gloo' = a "
                         "rehearsal of the N-rank code path on fewer GPUs than ranks (rank r uses GPU r %% #GPUs and the "
                         "collectives go through host memory) -- its numbers are not a scaling measurement")
    ap.add_argument("--hw", type=int, nargs=2, metavar=("H", "W"), default=None,
                    help="model workloads: override the input size (rehearsals and tests; the reported workload says so)")
    ap.add_argument("--amp", choices=("off", "bf16"), default="off",
                    help="model workloads: forward and loss under torch.autocast(bfloat16) -- fp32 parameters, 16-bit "
                         "activations, the DDCN layers on their native 16-bit forward / backward.  An extra figure: the "
                         "reference trains in fp32 and the BASELINE workloads are quoted in fp32 (the default)")
    ap.add_argument("--force-amp-r101", action="store_true",
                    help="run --workload r101 --amp bf16 at full size although that combination has faulted the GPU in "
                         "PyTorch's / rocBLAS's bf16 kernels of the DANet head (DESIGN.md section 6); not for shared machines")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-other-schedule", action="store_true",
                    help="layer: do not also time the step under the other backward schedule (for a rocprofv3 run whose "
                         "per-kernel averages should describe the default schedule only)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.workload == "r101" and args.amp != "off" and not args.hw and not args.force_amp_r101:
        raise SystemExit("bench.py: --workload r101 --amp bf16 at 769x769 batch 8 is refused: both runs of it ended in a GPU "
                         "memory fault, the second with every DDCN call on the fp32 kernels (the fault is raised inside the "
                         "DANet head's attention, after MIOpen's synchronous find of its last 1x1 convolution: PyTorch / "
                         "rocBLAS bf16 kernels on the 8 x 9409 x 9409 map; DESIGN.md section 6).  The BASELINE workloads are "
                         "fp32; --hw H W runs a reduced size, --force-amp-r101 overrides.")

    if os.environ.get("DDCN_BENCH_WATCHDOG"):     # debugging aid: dump every thread's stack after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["DDCN_BENCH_WATCHDOG"]), exit=False)
    launch = _load_launch()
    if not launch.under_launcher():
        if args.gpus > 1:
            # The N ranks are started here, by a parent that has made no GPU call (device_count() does not initialise
            # the GPU on this image); each is a fresh interpreter, never a re-exec of a process that touched the GPU.
            have = torch.cuda.device_count()
            if have < (args.gpus if args.backend == "nccl" else 1):
                raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible" % (args.gpus, have))
            sys.exit(launch.spawn_ranks(args.gpus, sys.argv[1:], script=os.path.abspath(__file__)))
        world, rank, local_rank = 1, 0, 0
    else:
        world = int(os.environ["WORLD_SIZE"])
        rank = int(os.environ["RANK"])
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if rank == 0 and args.gpus != world:
            print("bench: --gpus %d but the launcher started %d ranks; running %d" % (args.gpus, world, world),
                  file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if world > 1 and args.backend == "nccl":
        rank, world, dev = launch.init_dist("pytorch", backend="nccl")   # "nccl" is RCCL on ROCm
    else:
        if world > 1:
            rank, world, _ = launch.init_dist("pytorch", backend="gloo")
        dev = torch.device("cuda", local_rank % torch.cuda.device_count())
        torch.cuda.set_device(dev)
    pkg = importlib.import_module("semantic-segmentation_amd")
    dist_utils = importlib.import_module("semantic-segmentation_amd.dist_utils")

    if args.workload == "cfg5":
        inference_workload(args, world, rank, dev)
        if world > 1:
            dist.destroy_process_group()
        return
    if args.workload != "layer":
        model_workload(args, world, rank, dev)
        if world > 1:
            dist.destroy_process_group()
        return

    w = workload(args)
    x, weight, offset, mask, gout = make_inputs(w, dev, seed=1234 + rank)
    x.requires_grad_(True); weight.requires_grad_(True); offset.requires_grad_(True); mask.requires_grad_(True)

    def step():
        for t in (x, weight, offset, mask):
            t.grad = None
        out = pkg.modulated_deform_conv(x, offset, mask, weight, None, 1, w["pad"], w["dil"], 1, 1)
        out.backward(gout)
        if world > 1:   # data-parallel gradient exchange of the layer's parameter (SUM then /world)
            dist_utils.allreduce_grads([weight])

    # set-up, before the W warm-up steps: the first ~50 ms of GPU work in a fresh process run 2 % slow (clock ramp, the
    # caching allocator's first blocks), which would otherwise be charged to whichever schedule is timed first
    preheat = 0 if (world > 1 and args.backend == "gloo") else PREHEAT_STEPS   # the one-GPU gloo rehearsal stays short
    for _ in range(preheat):
        step()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = max_over_ranks(dt, dev)

    # the same step under the OTHER backward schedule (one stream <-> weight-gradient GEMM on a side stream beside the
    # scatter), so that the line shows what the default buys; per-kernel rooflines below are timed one kernel at a time,
    # which is what the one-stream schedule runs
    dcn = importlib.import_module("semantic-segmentation_amd.dcn.deform_conv")
    default_two = bool(dcn.TWO_STREAM_BACKWARD)
    alt = None
    if world == 1 and not args.no_other_schedule:
        dcn.TWO_STREAM_BACKWARD = not default_two
        for _ in range(max(1, args.warmup)):
            step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        alt = (time.perf_counter() - t1) / args.steps * 1e3
        dcn.TWO_STREAM_BACKWARD = default_two

    a = alg_numbers(w)
    images = w["N"] * world * args.steps
    devices = job_devices(dev, world)
    line = {
        "metric": "DDCN op fwd+bwd GB/s",
        "value": round(a["b_total"] * images / dt / 1e9, 2),
        "unit": "GB/s",
        "n_gpus": dist.get_world_size() if world > 1 else 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[0] layer: modulated DDCN 3x3, N=%d/GPU, C=%d->%d, %dx%d, "
                               "stride 1, pad=dil=%d, groups 1, dg 1, fwd+bwd" %
                               (w["N"], w["C"], w["Co"], w["H"], w["W"], w["dil"]),
                   "images_per_gpu": w["N"], "parallelism": "dp%d" % world, "preheat_steps": preheat,
                   "alg_bytes_per_image": a["b_total"], "alg_flops_per_image": a["f_total"],
                   "backward_schedule": "two streams (G3 beside the scatter)" if default_two else
                                        "one stream: every kernel runs alone, in order"},
        "op_tflops": round(a["f_total"] * images / dt / 1e12, 2),
        "images_per_s": round(images / dt, 2),
        "devices": devices,
    }
    if alt is not None:
        line["ms_per_step_other_schedule"] = {"schedule": "one stream" if default_two else "two streams",
                                              "ms_per_step": round(alt, 4)}
    if world > 1:
        # the layer's own exchange (grad_weight) and, for scale, the gradient of BASELINE configs[3]'s model
        # (DDCN-ResNet101 + DANet head: ~69 M fp32 parameters = 276 MB in one step)
        line["allreduce"] = {"grad_weight": allreduce_probe(dev, world, weight.numel() * 4),
                             "configs3_gradient": allreduce_probe(dev, world, 69_000_000 * 4, iters=5),
                             "backend": "nccl (RCCL)" if args.backend == "nccl" else "gloo (rehearsal)",
                             "distinct_devices": len({tuple(d[1:]) for d in devices})}
    if rank == 0 and world == 1 and not args.no_kernel_timing:
        ks, traffic_src = kernel_rooflines(w, [t.detach() for t in (x, weight, offset, mask)] + [gout],
                                           max(5, args.steps // 2))
        dom = max((k for k in ks if k["bound"] == "mfma"), key=lambda k: k["ms"])
        line["roofline"] = {k: dom[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
        line["roofline"]["kernel"] = dom["kernel"]
        line["roofline"]["traffic_source"] = traffic_src
        # the bytes the kernels that really ran moved, over their own time: im2col runs ONCE per step here (the
        # columns are kept for backward) and the scatter is one fused pass, while `value` prices the reference's
        # four passes (2 x im2col + col2im + col2im_coord) against the whole step
        gath = [k for k in ks if k["bound"] == "hbm"]
        line["gather_GBps_executed"] = round(sum(k["work"] for k in gath) / sum(k["ms"] for k in gath) / 1e6, 1)
        line["gather_ms_executed"] = round(sum(k["ms"] for k in gath), 4)
        for k in ks:
            del k["work"]
        line["roofline_kernels"] = ks
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(w)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
