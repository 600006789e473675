import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
os.environ["DDCN_GEMM_SK"] = "1"
_lib = importlib.import_module("semantic-segmentation_amd.dcn._lib")
L = _lib.lib()
DEV = "cuda:0"
N, Ci, Co, H, W = 1, 64, 160, 40, 45
g = _lib.Geom(N, Ci, H, W, Co, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, H, W)
P = H * W; Kg = Ci * 9; ld = L.ddcn_col_ld(g)
rng = np.random.default_rng(6)
w = rng.standard_normal((Co, Kg)).astype(np.float32)
gout = rng.standard_normal((N, Co, P)).astype(np.float32)
wd, gd = torch.from_numpy(w).to(DEV), torch.from_numpy(gout).to(DEV)
gcol = torch.full((N, Kg, ld), float("nan"), device=DEV)
s = _lib.current_stream(torch.device(DEV))
_lib.check(L.ddcn_gemm_gcol_f32(_lib.ptr(wd), _lib.ptr(gd), _lib.ptr(gcol), g, s))
torch.cuda.synchronize()
got = gcol[0, :, :P].cpu().numpy()
exp = (w.astype(np.float64).T @ gout[0].astype(np.float64))
bad = ~(np.abs(got - exp) <= 1e-3)
print("shape", got.shape, "bad frac", bad.mean(), "nan frac", np.isnan(got).mean(), "zero frac", (got == 0).mean())
# per 128x128 tile bad fraction
for tm in range((Kg + 127) // 128):
    print(" ".join("%4.2f" % bad[tm*128:(tm+1)*128, tn*128:(tn+1)*128].mean() for tn in range((P + 127) // 128)))
# inside tile (0,0): per row bad
print("tile00 rows bad:", [int(bad[r, :128].any()) for r in range(0, 128)])
print("tile00 cols bad:", [int(bad[:128, c].any()) for c in range(0, 128)])
r = got[:8, :4]; print(r); print(exp[:8, :4])
