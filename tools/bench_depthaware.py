"""Depth-aware convolution (K:868-996) on the bench layer's shape: stage kernels alone and the op through autograd.
python tools/bench_depthaware.py [N C H W dil]"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

pkg = importlib.import_module("semantic-segmentation_amd")
_lib = importlib.import_module("semantic-segmentation_amd.dcn._lib")
L = _lib.lib()
dev = torch.device("cuda:0")
N, C, H, W, D = (int(v) for v in sys.argv[1:6]) if len(sys.argv) >= 6 else (2, 512, 97, 97, 4)
torch.manual_seed(0)
x = torch.randn(N, C, H, W, device=dev)
depth = torch.rand(N, 1, H, W, device=dev) * 4
w = (torch.rand(C, C, 3, 3, device=dev) * 2 - 1) / (C * 9) ** 0.5
gout = torch.randn(N, C, H, W, device=dev)
g = _lib.Geom(N, C, H, W, C, 3, 3, 1, 1, D, D, D, D, 1, 1, H, W)
nb = L.ddcn_workspace_bytes(g)
cols = torch.empty(nb // 4, device=dev)
gx = torch.zeros_like(x)
p = _lib.ptr
s = _lib.current_stream(dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


P = H * W
col_bytes = N * C * 9 * P * 4
x_bytes = N * C * H * W * 4
t = timed(lambda: _lib.check(L.ddcn_depthaware_im2col_f32(p(x), p(depth), p(cols), g, s)))
print("depthaware_im2col   %.4f ms  %7.1f GB/s (x read + columns written)" % (t, (col_bytes + x_bytes) / t / 1e6))
t = timed(lambda: _lib.check(L.ddcn_depthaware_col2im_f32(p(cols), p(depth), p(gx), g, s)))
print("depthaware_col2im   %.4f ms  %7.1f GB/s (columns read + grad_input read-modify-write)" % (t, (col_bytes + 2 * x_bytes) / t / 1e6))
xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)


def step():
    xg.grad = None
    wg.grad = None
    pkg.depthaware_conv(xg, depth, wg, None, 1, D, D, 1).backward(gout)


print("DepthawareConv fwd+bwd through autograd  %.3f ms" % timed(step))
