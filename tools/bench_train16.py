"""fwd+bwd of the bench layer on bfloat16 tensors (training): native 16-bit forward + 16-bit G2 vs all-fp32 kernels.
python tools/bench_train16.py"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

pkg = importlib.import_module("semantic-segmentation_amd")
dcn = importlib.import_module("semantic-segmentation_amd.dcn.deform_conv")
dev = torch.device("cuda:0")
torch.manual_seed(0)
N, C, H, D = 2, 512, 97, 4
for dt in (torch.bfloat16, torch.float16):
    x = torch.randn(N, C, H, H, device=dev).to(dt).requires_grad_(True)
    w = ((torch.rand(C, C, 3, 3, device=dev) * 2 - 1) / (C * 9) ** 0.5).to(dt).requires_grad_(True)
    off = (torch.randn(N, 18, H, H, device=dev) * 2).to(dt).requires_grad_(True)
    mask = torch.sigmoid(torch.randn(N, 9, H, H, device=dev)).to(dt).requires_grad_(True)
    gout = torch.randn(N, C, H, H, device=dev).to(dt)
    for native in (True, False):
        dcn.TRAIN16_NATIVE = native

        def step():
            for t in (x, w, off, mask):
                t.grad = None
            pkg.modulated_deform_conv(x, off, mask, w, None, 1, D, D, 1, 1).backward(gout)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            step()
        b.record()
        b.synchronize()
        print("%-9s %-40s %.3f ms per fwd+bwd" % (str(dt).split(".")[1], "native fwd, 16-bit G2 and G3" if native else
                                                  "fp32 kernels (convert in / out)", a.elapsed_time(b) / 20), flush=True)
