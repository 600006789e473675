"""Times the native bf16 forward on BASELINE config 5's layer shapes (ModulatedDeformConvPack, dg=4, 1024x2048 input).

Per layer: the whole call (im2col_bf16 + gemm_bf16) with HIP events; rocprofv3 --kernel-trace --stats of this
script gives the split.  Algorithmic bytes im2col = 2*(C*H*W + 27*dg*P + C*9*P); flops = 2*Co*C*9*P.
usage: python tools/bench_bf16.py [iters]     (DDCN_BENCH_SIGMA = standard deviation of the offsets in pixels, default 2)
"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_lib = importlib.import_module("semantic-segmentation_amd.dcn._lib")

LAYERS = [  # layer1 at 1/4 resolution, the dilated stages at 1/8
    ("layer1 C64 256x512 d1", 64, 256, 512, 1),
    ("layer2 C128 128x256 d1", 128, 128, 256, 1),
    ("layer3 C256 128x256 d2", 256, 128, 256, 2),
    ("layer4 C512 128x256 d4", 512, 128, 256, 4),
]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev, dg = "cuda:0", 4
    sigma = float(os.environ.get("DDCN_BENCH_SIGMA", 2))
    L, p = _lib.lib(), _lib.ptr
    for name, C, H, W, d in LAYERS:
        g = _lib.Geom(1, C, H, W, C, 3, 3, 1, 1, d, d, d, d, 1, dg, H, W)
        assert L.ddcn_bf16_supported(g) == 0
        P = H * W
        x = torch.randn(1, C, H, W, device=dev).bfloat16()
        w = (torch.randn(C, C, 3, 3, device=dev) / (C * 9) ** 0.5).bfloat16()
        off = (sigma * torch.randn(1, dg * 18, H, W, device=dev)).bfloat16()
        mask = torch.sigmoid(torch.randn(1, dg * 9, H, W, device=dev)).bfloat16()
        out = torch.empty(1, C, H, W, device=dev, dtype=torch.bfloat16)
        nbytes = L.ddcn_workspace_bytes_bf16(g)
        ws = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=dev)
        s = _lib.current_stream(x.device)

        def run():
            _lib.check(L.ddcn_modulated_deform_conv_forward_bf16(p(x), p(w), None, p(off), p(mask), p(out), p(ws),
                                                                 nbytes, g, s))

        def timed(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters
        ms = timed(run)
        flops = 2.0 * C * C * 9 * P
        gbytes = 2.0 * (C * H * W + 27 * dg * P + C * 9 * P)
        print("%-26s %.3f ms   GEMM %.1f GFLOP (%.0f TFLOP/s if alone)   im2col %.0f MB (%.0f GB/s if alone)"
              % (name, ms, flops / 1e9, flops / ms / 1e9, gbytes / 1e6, gbytes / ms / 1e6))
        # the same layer without a columns buffer (ddcn_fused16.hip)
        if L.ddcn_fused16_supported(g) == 0:
            fbytes = L.ddcn_fused16_workspace_bytes(g)
            fws = torch.empty(fbytes // 2, dtype=torch.bfloat16, device=dev)
            out2 = torch.empty_like(out)

            def run_fused():
                _lib.check(L.ddcn_modulated_deform_conv_forward_fused_bf16(p(x), p(w), None, p(off), p(mask), p(out2),
                                                                           p(fws), fbytes, g, s))
            msf = timed(run_fused)
            run()
            torch.cuda.synchronize()
            diff = float((out.float() - out2.float()).abs().max() / out.float().abs().max())
            print("%-26s %.3f ms   fused, no columns: %.0f TFLOP/s for the whole call; max |fused - columns| / max = %.1e"
                  % ("", msf, flops / msf / 1e9, diff))


if __name__ == "__main__":
    main()
