"""Times deformable PS-RoI pooling (forward, backward) on one GPU through the C ABI.

Workload: a detection head -- data[B,C,H,W] stride-16 features, R RoIs, 7x7 bins, 4x4 samples per bin.
Algorithmic HBM bytes: forward = data read once + out and out_count written; backward = grad_out and out_count
read + data read once + grad_data written once (+ offset tensors, negligible).
usage: python tools/bench_psroi.py [key=value ...]   (B C H W R out spp gs)
"""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_lib = importlib.import_module("semantic-segmentation_amd.dcn._lib")


def main():
    w = dict(B=2, C=256, H=50, W=68, R=1024, out=7, spp=4, gs=1, iters=20)
    for a in sys.argv[1:]:
        k, v = a.split("=")
        w[k] = int(v)
    dev = "cuda:0"
    rng = np.random.default_rng(0)
    oc = w["C"] // (w["gs"] ** 2)
    data = torch.randn(w["B"], w["C"], w["H"], w["W"], device=dev)
    iw, ih = w["W"] * 16, w["H"] * 16
    x1, y1 = rng.uniform(0, 0.7 * iw, w["R"]), rng.uniform(0, 0.7 * ih, w["R"])
    bw, bh = rng.uniform(32, 0.5 * iw, w["R"]), rng.uniform(32, 0.5 * ih, w["R"])
    rois = np.stack([rng.integers(0, w["B"], w["R"]), x1, y1, np.minimum(x1 + bw, iw - 1), np.minimum(y1 + bh, ih - 1)], 1)
    rois = torch.from_numpy(rois.astype(np.float32)).to(dev)
    offset = 0.5 * torch.randn(w["R"], 2, w["out"], w["out"], device=dev)
    out = torch.empty(w["R"], oc, w["out"], w["out"], device=dev)
    cnt = torch.empty_like(out)
    gout = torch.randn_like(out)
    gdata = torch.zeros_like(data)
    goff = torch.zeros_like(offset)
    g = _lib.RoiGeom(w["B"], w["C"], w["H"], w["W"], w["R"], oc, w["out"], w["gs"], w["out"], w["spp"], 2, 0,
                     1 / 16, 0.1)
    L, p, s = _lib.lib(), _lib.ptr, _lib.current_stream(data.device)

    def fwd():
        _lib.check(L.ddcn_deform_psroi_pool_forward_f32(p(data), p(rois), p(offset), p(out), p(cnt), g, s))

    def bwd():
        _lib.check(L.ddcn_deform_psroi_pool_backward_f32(p(gout), p(data), p(rois), p(offset), p(cnt), p(gdata),
                                                         p(goff), g, s))

    res = {}
    for name, fn, nbytes in (("forward", fwd, data.numel() * 4 + 2 * out.numel() * 4),
                             ("backward", bwd, 2 * out.numel() * 4 + 2 * data.numel() * 4)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(w["iters"]):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / w["iters"]
        samples = out.numel() * w["spp"] ** 2
        res[name] = ms
        print("%-8s %.3f ms   %.1f G samples/s   algorithmic %.1f MB -> %.0f GB/s (%.1f%% of 8 TB/s)"
              % (name, ms, samples / ms / 1e6, nbytes / 1e6, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80))
    print(w)


if __name__ == "__main__":
    main()
