"""Experiment: backward of the bench layer with the weight-gradient GEMM (G3) on a side stream beside the scatter.
G2 -> scatter on the main stream; G3 (needs only grad_out and the saved columns) on a second stream.
python tools/bench_overlap.py [order]   order: seq | g3_beside_scatter | g3_beside_all"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

_lib = importlib.import_module("semantic-segmentation_amd.dcn._lib")
L = _lib.lib()
dev = torch.device("cuda:0")
w = dict(N=2, C=512, Co=512, H=97, W=97, k=3, pad=4, dil=4, dg=1)
x, weight, offset, mask, gout = bench.make_inputs(w, dev, 0)
g = _lib.Geom(2, 512, 97, 97, 512, 3, 3, 1, 1, 4, 4, 4, 4, 1, 1, 97, 97)
nb = L.ddcn_workspace_bytes(g)
cols = torch.empty(nb // 4, device=dev)
gcol = torch.empty(nb // 4, device=dev)
gx, goff, gmask, gw = (torch.zeros_like(t) for t in (x, offset, mask, weight))
p = _lib.ptr
main = torch.cuda.current_stream()
side = torch.cuda.Stream(priority=0)
sm = _lib.current_stream(dev)
ss = __import__("ctypes").c_void_p(side.cuda_stream)
_lib.check(L.ddcn_im2col_f32(p(x), p(offset), p(mask), p(cols), g, sm))


def seq():
    _lib.check(L.ddcn_gemm_gcol_f32(p(weight), p(gout), p(gcol), g, sm))
    _lib.check(L.ddcn_col2im_coord_f32(p(gcol), p(x), p(offset), p(mask), p(gx), p(goff), p(gmask), g, sm))
    _lib.check(L.ddcn_gemm_gweight_f32(p(gout), p(cols), p(gw), g, sm))


def g3_beside_scatter():
    _lib.check(L.ddcn_gemm_gcol_f32(p(weight), p(gout), p(gcol), g, sm))
    ev = torch.cuda.Event()
    ev.record(main)
    side.wait_event(ev)
    _lib.check(L.ddcn_gemm_gweight_f32(p(gout), p(cols), p(gw), g, ss))
    _lib.check(L.ddcn_col2im_coord_f32(p(gcol), p(x), p(offset), p(mask), p(gx), p(goff), p(gmask), g, sm))
    ev2 = torch.cuda.Event()
    ev2.record(side)
    main.wait_event(ev2)


def scatter_first_then_g3_side():
    _lib.check(L.ddcn_gemm_gcol_f32(p(weight), p(gout), p(gcol), g, sm))
    ev = torch.cuda.Event()
    ev.record(main)
    side.wait_event(ev)
    _lib.check(L.ddcn_col2im_coord_f32(p(gcol), p(x), p(offset), p(mask), p(gx), p(goff), p(gmask), g, ss))
    _lib.check(L.ddcn_gemm_gweight_f32(p(gout), p(cols), p(gw), g, sm))
    ev2 = torch.cuda.Event()
    ev2.record(side)
    main.wait_event(ev2)


g1 = _lib.Geom(1, 512, 97, 97, 512, 3, 3, 1, 1, 4, 4, 4, 4, 1, 1, 97, 97)
import ctypes


def off_ptr(t, n):
    return ctypes.c_void_p(t.data_ptr() + n * t[0].numel() * 4)


gcol_img = gcol.view(2, -1)


def pipelined():
    """G2 per image on the main stream; the scatter of image n on the side stream beside G2 of image n+1 / G3."""
    evs = []
    for n in range(2):
        _lib.check(L.ddcn_gemm_gcol_f32(p(weight), off_ptr(gout, n), off_ptr(gcol_img, n), g1, sm))
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        _lib.check(L.ddcn_col2im_coord_f32(off_ptr(gcol_img, n), off_ptr(x, n), off_ptr(offset, n), off_ptr(mask, n),
                                           off_ptr(gx, n), off_ptr(goff, n), off_ptr(gmask, n), g1, ss))
    _lib.check(L.ddcn_gemm_gweight_f32(p(gout), p(cols), p(gw), g, sm))
    ev2 = torch.cuda.Event()
    ev2.record(side)
    main.wait_event(ev2)


out = torch.empty(2, 512, 97, 97, device=dev)
cols_img = cols.view(2, -1)


def fwd_seq():
    _lib.check(L.ddcn_im2col_f32(p(x), p(offset), p(mask), p(cols), g, sm))
    _lib.check(L.ddcn_gemm_out_f32(p(weight), p(cols), None, p(out), g, sm))


def fwd_pipelined():
    """im2col of image n+1 on the side stream beside G1 of image n."""
    evs = []
    ev0 = torch.cuda.Event()
    ev0.record(main)
    side.wait_event(ev0)
    for n in range(2):
        _lib.check(L.ddcn_im2col_f32(off_ptr(x, n), off_ptr(offset, n), off_ptr(mask, n), off_ptr(cols_img, n), g1, ss))
        ev = torch.cuda.Event()
        ev.record(side)
        evs.append(ev)
    for n in range(2):
        main.wait_event(evs[n])
        _lib.check(L.ddcn_gemm_out_f32(p(weight), off_ptr(cols_img, n), None, off_ptr(out, n), g1, sm))


for name, fn in (("forward sequential", fwd_seq), ("forward, im2col(n+1) beside G1(n)", fwd_pipelined), ("sequential", seq), ("pipelined per image", pipelined), ("G3 on side stream beside scatter", g3_beside_scatter),
                 ("scatter on side stream beside G3", scatter_first_then_g3_side)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30):
        fn()
    b.record()
    torch.cuda.synchronize()
    print("%-40s %.4f ms per backward" % (name, a.elapsed_time(b) / 30), flush=True)
