"""Mask-branch (DeformConv depth C/2 -> 9, res5 shape) direct forward / backward alone, for profiling:
python tools/bench_direct.py [iters]"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

_lib = importlib.import_module("semantic-segmentation_amd.dcn._lib")
L = _lib.lib()
dev = torch.device("cuda:0")
N, C, H, D, CO = 2, 256, 97, 4, 9
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = _lib.Geom(N, C, H, H, CO, 3, 3, 1, 1, D, D, D, D, 1, 1, H, H)
gen = torch.Generator().manual_seed(0)
x = torch.randn(N, C, H, H, generator=gen).to(dev)
w = (torch.randn(CO, C, 3, 3, generator=gen) * 0.02).to(dev)
off = (torch.randn(N, 18, H, H, generator=gen) * 2).to(dev)
gout = torch.randn(N, CO, H, H, generator=gen).to(dev)
out = torch.empty_like(gout)
gx, goff, gw = torch.zeros_like(x), torch.zeros_like(off), torch.zeros_like(w)
nbytes = L.ddcn_workspace_bytes(g)
ws = torch.empty(nbytes // 4, device=dev)
s = _lib.current_stream(dev)
p = _lib.ptr


def fwd():
    _lib.check(L.ddcn_deform_conv_direct_forward_f32(p(x), p(w), None, p(off), p(out), 1, p(ws), nbytes, g, s))


def bwd():
    _lib.check(L.ddcn_deform_conv_direct_backward_f32(p(x), p(w), p(off), p(gout), p(gx), p(goff), p(gw), None, p(ws),
                                                      nbytes, g, s))


for name, fn in (("direct forward", fwd), ("direct backward", bwd)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    print("%-18s %.4f ms" % (name, a.elapsed_time(b) / iters), flush=True)
