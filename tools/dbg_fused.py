import importlib, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
_lib = importlib.import_module("semantic-segmentation_amd.dcn._lib")
from oracle import ddcn_oracle as orc
L, p = _lib.lib(), _lib.ptr
dev = "cuda:0"
C, H, W, d, dg = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
sigma = float(sys.argv[6]) if len(sys.argv) > 6 else 2.0
torch.manual_seed(0)
g = _lib.Geom(1, C, H, W, C, 3, 3, 1, 1, d, d, d, d, 1, dg, H, W)
x = torch.randn(1, C, H, W, device=dev).bfloat16()
w = (torch.randn(C, C, 3, 3, device=dev) / (C * 9) ** 0.5).bfloat16()
off = (sigma * torch.randn(1, dg * 18, H, W, device=dev)).bfloat16()
mask = torch.sigmoid(torch.randn(1, dg * 9, H, W, device=dev)).bfloat16()
s = _lib.current_stream(x.device)
outs = []
for fused in (False, True, True):
    out = torch.full((1, C, H, W), float("nan"), device=dev, dtype=torch.bfloat16)
    if fused:
        nb = L.ddcn_fused16_workspace_bytes(g); ws = torch.empty(nb // 2, dtype=torch.bfloat16, device=dev)
        _lib.check(L.ddcn_modulated_deform_conv_forward_fused_bf16(p(x), p(w), None, p(off), p(mask), p(out), p(ws), nb, g, s))
    else:
        nb = L.ddcn_workspace_bytes_bf16(g); ws = torch.empty(nb // 2, dtype=torch.bfloat16, device=dev)
        _lib.check(L.ddcn_modulated_deform_conv_forward_bf16(p(x), p(w), None, p(off), p(mask), p(out), p(ws), nb, g, s))
    torch.cuda.synchronize()
    outs.append(out.float().cpu().numpy()[0])
col, f1, f2 = outs
print("fused run-to-run identical:", np.array_equal(f1, f2), "nan:", np.isnan(f1).sum())
f64 = lambda t: t.double().cpu().numpy()
exp = orc.deform_forward(f64(x), f64(off), f64(mask), f64(w), None, 1, d, d, 1, dg)[0]
sc = np.abs(exp).max()
for name, o in (("columns", col), ("fused", f1)):
    e = np.abs(o - exp) / sc
    print(name, "max err/scale %.2e" % e.max(), "frac > 1e-2: %.2e" % (e > 1e-2).mean())
e = np.abs(f1 - exp) / sc
bad = e > 1e-2
print("bad by channel (first 16):", bad.reshape(C, -1).mean(1)[:16].round(3))
bp = bad.any(0)
print("bad pixels frac %.4f" % bp.mean())
ys, xs = np.nonzero(bp)
if len(ys):
    print("bad pixel rows hist (by tile row of 8):", np.bincount(ys // 8, minlength=H // 8)[:20])
    print("bad pixel cols hist (by tile col of 16):", np.bincount(xs // 16, minlength=W // 16)[:20])
    print("in-tile y:", np.bincount(ys % 8, minlength=8), "in-tile x:", np.bincount(xs % 16, minlength=16))
