"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Bar (BASELINE.json north_star): fp32 forward rel-err <= 1e-5, backward <= 1e-4 (atomic reordering), with
rel-err := max|a-b| / max|b| per tensor against the fp64 oracle evaluated on the same fp32-rounded inputs.
"""
import ctypes
import importlib

import numpy as np
import pytest
import torch

import cases
from helpers import BWD_TOL, FWD_TOL, load_golden
from oracle import ddcn_oracle as orc

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    pkg = importlib.import_module("semantic-segmentation_amd")
    _lib = importlib.import_module("semantic-segmentation_amd.dcn._lib")
DEV = "cuda:0"


def _t(a, grad=False):
    if a is None:
        return None
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
    return t.requires_grad_(True) if grad else t


def _r32(inp):
    return {k: (None if v is None else v.astype(np.float32).astype(np.float64)) for k, v in inp.items()}


def _run_hip(c, inp):
    x, off, w = _t(inp["x"], True), _t(inp["offset"], True), _t(inp["weight"], True)
    m = _t(inp.get("mask"), True) if c["modulated"] else None
    b = _t(inp.get("bias"), True) if c["bias"] else None
    if c["modulated"]:
        out = pkg.modulated_deform_conv(x, off, m, w, b, c["stride"], c["pad"], c["dil"], c["groups"], c["dg"])
    else:
        out = pkg.deform_conv(x, off, w, b, c["stride"], c["pad"], c["dil"], c["groups"], c["dg"])
    out.backward(_t(inp["gout"]))
    torch.cuda.synchronize()
    g = lambda t: None if t is None else t.grad.cpu().numpy()
    return out.detach().cpu().numpy(), dict(gx=g(x), goff=g(off), gmask=g(m), gw=g(w), gb=g(b))


def _assert_case(c, out, grads, exp):
    assert cases.relerr(out, exp["out"]) <= FWD_TOL, ("out", cases.relerr(out, exp["out"]))
    for k in ("gx", "goff", "gw"):
        assert cases.relerr(grads[k], exp[k]) <= BWD_TOL, (k, cases.relerr(grads[k], exp[k]))
    if c["modulated"]:
        assert cases.relerr(grads["gmask"], exp["gmask"]) <= BWD_TOL, "gmask"
    if c["bias"]:
        assert cases.relerr(grads["gb"], exp["gb"]) <= BWD_TOL, "gb"


def test_extension_is_the_hip_library():
    assert _lib.lib().ddcn_version() == 100
    assert "libddcn_hip.so" in _lib.LIB_PATH


@pytest.mark.parametrize("c", cases.SMALL_CASES, ids=lambda c: c["name"])
def test_hip_matches_golden_fixtures(c):
    g = load_golden(c["name"])
    out, grads = _run_hip(c, g)
    _assert_case(c, out, grads, g)


@pytest.mark.parametrize("c", cases.MEDIUM_CASES, ids=lambda c: c["name"])
def test_hip_matches_oracle_medium(c):
    inp = _r32(cases.make_inputs(c))
    kw = cases.kwargs(c)
    exp = dict(out=orc.deform_forward(inp["x"], inp["offset"], inp["mask"], inp["weight"], inp["bias"], **kw))
    exp["gx"], exp["goff"], exp["gmask"], exp["gw"], exp["gb"] = orc.deform_backward(
        inp["x"], inp["offset"], inp["mask"], inp["weight"], inp["gout"], c["bias"], **kw)
    out, grads = _run_hip(c, inp)
    _assert_case(c, out, grads, exp)


@pytest.mark.parametrize("c", cases.DEPTHAWARE_CASES, ids=lambda c: c["name"])
def test_hip_depthaware_matches_golden(c):
    g = load_golden(c["name"])
    x, w = _t(g["x"], True), _t(g["weight"], True)
    b = _t(g.get("bias"), True) if c["bias"] else None
    out = pkg.depthaware_conv(x, _t(g["depth"]), w, b, c["stride"], c["pad"], c["dil"], c["groups"])
    out.backward(_t(g["gout"]))
    assert cases.relerr(out.detach().cpu().numpy(), g["out"]) <= FWD_TOL
    assert cases.relerr(x.grad.cpu().numpy(), g["gx"]) <= BWD_TOL
    assert cases.relerr(w.grad.cpu().numpy(), g["gw"]) <= BWD_TOL
    if c["bias"]:
        assert cases.relerr(b.grad.cpu().numpy(), g["gb"]) <= BWD_TOL


DA_MEDIUM = [
    cases.C("da_med_c64_33x29_d2", N=2, Ci=64, Co=48, H=33, W=29, pad=2, dil=2, bias=True, seed=45),
    cases.C("da_med_c20_41x37_s2", N=2, Ci=20, Co=24, H=41, W=37, stride=2, pad=1, dil=1, seed=46),   # C % 8 != 0
    cases.C("da_med_c16_30x30_s2_d3_g2", N=1, Ci=16, Co=16, H=30, W=30, stride=2, pad=4, dil=3, groups=2, seed=47),
]


@pytest.mark.parametrize("c", DA_MEDIUM, ids=lambda c: c["name"])
def test_hip_depthaware_medium_against_oracle(c):
    """Depth-aware conv at sizes that span many workgroups; grad_input comes from the gather formulation of K:952-996
    (one lane per input cell, no atomics), so a second run must give the same bits."""
    inp = _r32(cases.make_inputs(c))
    kw = dict(stride=c["stride"], padding=c["pad"], dilation=c["dil"], groups=c["groups"])
    eout = orc.depthaware_forward(inp["x"], inp["depth"], inp["weight"], inp["bias"], **kw)
    egx, egw, egb = orc.depthaware_backward(inp["x"], inp["depth"], inp["weight"], inp["gout"], c["bias"], **kw)

    def run():
        x, w = _t(inp["x"], True), _t(inp["weight"], True)
        b = _t(inp["bias"], True) if c["bias"] else None
        out = pkg.depthaware_conv(x, _t(inp["depth"]), w, b, c["stride"], c["pad"], c["dil"], c["groups"])
        out.backward(_t(inp["gout"]))
        return out.detach(), x.grad, w.grad, (b.grad if c["bias"] else None)
    out, gx, gw, gb = run()
    assert cases.relerr(out.cpu().numpy(), eout) <= FWD_TOL
    assert cases.relerr(gx.cpu().numpy(), egx) <= BWD_TOL
    assert cases.relerr(gw.cpu().numpy(), egw) <= BWD_TOL
    if c["bias"]:
        assert cases.relerr(gb.cpu().numpy(), egb) <= BWD_TOL
    assert torch.equal(run()[1], gx)          # deterministic grad_input


# ---------------------------------------------------------------- stage level, straight through the C ABI
def _geom(c):
    Ho, Wo, _ = cases.geometry(c)
    return _lib.Geom(c["N"], c["Ci"], c["H"], c["W"], c["Co"], c["kh"], c["kw"], c["stride"], c["stride"], c["pad"],
                     c["pad"], c["dil"], c["dil"], c["groups"], c["dg"], Ho, Wo)


@pytest.mark.parametrize("c", [cases.SMALL_CASES[3], cases.SMALL_CASES[12], cases.MEDIUM_CASES[0],
                               cases.MEDIUM_CASES[2]], ids=lambda c: c["name"])
def test_stage_im2col_and_scatter_against_oracle(c):
    inp = _r32(cases.make_inputs(c))
    g = _geom(c)
    L = _lib.lib()
    K2, P = c["kh"] * c["kw"], g.Ho * g.Wo
    ld = L.ddcn_col_ld(g)
    x, off, msk = _t(inp["x"]), _t(inp["offset"]), _t(inp["mask"])
    col = torch.full((c["N"], c["Ci"] * K2, ld), float("nan"), device=DEV)
    s = _lib.current_stream(torch.device(DEV))
    _lib.check(L.ddcn_im2col_f32(_lib.ptr(x), _lib.ptr(off), _lib.ptr(msk), _lib.ptr(col), g, s))
    torch.cuda.synchronize()
    got = col[:, :, :P].cpu().numpy()
    for n in range(c["N"]):
        exp = orc.im2col(inp["x"][n], inp["offset"][n], inp["mask"][n], c["kh"], c["kw"], c["stride"], c["pad"],
                         c["dil"], c["dg"])
        assert cases.relerr(got[n], exp) <= FWD_TOL
    # scatter stage: feed a random gcol
    rng = np.random.default_rng(99)
    gcol = rng.standard_normal((c["N"], c["Ci"] * K2, P)).astype(np.float32)
    gcol_d = torch.zeros((c["N"], c["Ci"] * K2, ld), device=DEV)
    gcol_d[:, :, :P] = torch.from_numpy(gcol).to(DEV)
    gx = torch.zeros_like(x)
    goff = torch.full_like(off, 7.0)     # must be overwritten
    gmask = torch.full_like(msk, 7.0)
    _lib.check(L.ddcn_col2im_coord_f32(_lib.ptr(gcol_d), _lib.ptr(x), _lib.ptr(off), _lib.ptr(msk), _lib.ptr(gx),
                                       _lib.ptr(goff), _lib.ptr(gmask), g, s))
    torch.cuda.synchronize()
    for n in range(c["N"]):
        egx = orc.col2im(gcol[n].astype(np.float64), inp["offset"][n], inp["mask"][n], c["Ci"], c["H"], c["W"],
                         c["kh"], c["kw"], c["stride"], c["pad"], c["dil"], c["dg"])
        egoff, egmask = orc.col2im_coord(gcol[n].astype(np.float64), inp["x"][n], inp["offset"][n], inp["mask"][n],
                                         c["kh"], c["kw"], c["stride"], c["pad"], c["dil"], c["dg"])
        assert cases.relerr(gx[n].cpu().numpy(), egx) <= BWD_TOL
        assert cases.relerr(goff[n].cpu().numpy(), egoff) <= BWD_TOL
        assert cases.relerr(gmask[n].cpu().numpy(), egmask) <= BWD_TOL


GEMM_CASES = [
    cases.MEDIUM_CASES[2],   # Ci=128, Co=136 (M edge), P=625 (N edge), groups 1
    # more column tiles than one round of workgroups: wide tiles + an ODD number of narrow tail tiles (one idle wave
    # group), K = 36 and K = 72 (partial K tiles on both operand layouts), split-K over 66049 pixels
    cases.C("gemm_tail_c4_257x257", N=1, Ci=4, Co=72, H=257, W=257, pad=1, dil=1, seed=60),
    # 3 row tiles x 2 images: odd tile counts in the wide class
    cases.C("gemm_odd_c8_150x150", N=2, Ci=8, Co=300, H=150, W=150, pad=1, dil=1, seed=61),
    # few output channels (the DDCN mask branch): the backward contractions run as bandwidth kernels (ddcn_skinny.hip);
    # K = 360 and K = 63 leave partial row blocks, 97 x 97 pixels split over several workgroups
    cases.C("skinny_co9_c40_97x97", N=3, Ci=40, Co=9, H=97, W=97, pad=2, dil=2, seed=62),
    cases.C("skinny_co12_c7_33x21", N=2, Ci=7, Co=12, H=33, W=21, pad=1, dil=1, seed=63),
]


@pytest.mark.parametrize("c", GEMM_CASES, ids=lambda c: c["name"])
def test_stage_gemms_against_fp64_matmul(c):
    g = _geom(c)
    L = _lib.lib()
    K2, P = 9, g.Ho * g.Wo
    K = c["Ci"] * K2
    ld = L.ddcn_col_ld(g)
    rng = np.random.default_rng(5)
    w = rng.standard_normal((c["Co"], K)).astype(np.float32)
    col = rng.standard_normal((c["N"], K, P)).astype(np.float32)
    gout = rng.standard_normal((c["N"], c["Co"], P)).astype(np.float32)
    bias = rng.standard_normal((c["Co"],)).astype(np.float32)
    col_d = torch.full((c["N"], K, ld), float("nan"), device=DEV)
    col_d[:, :, :P] = torch.from_numpy(col).to(DEV)
    w_d, gout_d, bias_d = _t(w), _t(gout), _t(bias)
    s = _lib.current_stream(torch.device(DEV))
    out = torch.empty((c["N"], c["Co"], P), device=DEV)
    _lib.check(L.ddcn_gemm_out_f32(_lib.ptr(w_d), _lib.ptr(col_d), _lib.ptr(bias_d), _lib.ptr(out), g, s))
    exp = np.einsum("ok,nkp->nop", w.astype(np.float64), col.astype(np.float64)) + bias.reshape(1, -1, 1)
    assert cases.relerr(out.cpu().numpy(), exp) <= FWD_TOL
    gcol = torch.full((c["N"], K, ld), float("nan"), device=DEV)
    _lib.check(L.ddcn_gemm_gcol_f32(_lib.ptr(w_d), _lib.ptr(gout_d), _lib.ptr(gcol), g, s))
    exp = np.einsum("ok,nop->nkp", w.astype(np.float64), gout.astype(np.float64))
    assert cases.relerr(gcol[:, :, :P].cpu().numpy(), exp) <= FWD_TOL
    gw = torch.ones((c["Co"], K), device=DEV)     # accumulate semantics: starts at 1
    _lib.check(L.ddcn_gemm_gweight_f32(_lib.ptr(gout_d), _lib.ptr(col_d), _lib.ptr(gw), g, s))
    exp = 1.0 + np.einsum("nop,nkp->ok", gout.astype(np.float64), col.astype(np.float64))
    assert cases.relerr(gw.cpu().numpy(), exp) <= BWD_TOL
    gb = torch.ones((c["Co"],), device=DEV)
    _lib.check(L.ddcn_bias_grad_f32(_lib.ptr(gout_d), _lib.ptr(gb), g, s))
    assert cases.relerr(gb.cpu().numpy(), 1.0 + gout.astype(np.float64).sum(axis=(0, 2))) <= BWD_TOL


@pytest.mark.parametrize("c", [cases.MEDIUM_CASES[0], cases.MEDIUM_CASES[1]], ids=lambda c: c["name"])
def test_backward_recompute_path_matches_saved_columns_path(c, monkeypatch):
    """The Python layer keeps the forward columns by default (ddcn_*_backward_saved_f32); with SAVE_COLUMNS off it
    calls the plain entry points, which re-run im2col like the reference (deform_conv_cuda.cpp:325-327).  Both must
    meet the oracle bar."""
    mod = importlib.import_module("semantic-segmentation_amd.dcn.deform_conv")
    inp = _r32(cases.make_inputs(c))
    kw = cases.kwargs(c)
    exp = dict(out=orc.deform_forward(inp["x"], inp["offset"], inp["mask"], inp["weight"], inp["bias"], **kw))
    exp["gx"], exp["goff"], exp["gmask"], exp["gw"], exp["gb"] = orc.deform_backward(
        inp["x"], inp["offset"], inp["mask"], inp["weight"], inp["gout"], c["bias"], **kw)
    res = {}
    for flag in (True, False):
        monkeypatch.setattr(mod, "SAVE_COLUMNS", flag)
        res[flag] = _run_hip(c, inp)
    for k, v in res[True][1].items():
        if v is not None:
            assert cases.relerr(res[False][1][k], v.astype(np.float64)) <= BWD_TOL, k
    assert np.array_equal(res[True][0], res[False][0])
    _assert_case(c, res[False][0], res[False][1], exp)


@pytest.mark.parametrize("geo", [
    dict(Ci=32, Co=9, H=40, W=37, k=3, dil=2, off=2.0),     # the direct kernel (mask branch shape)
    dict(Ci=16, Co=9, H=23, W=150, k=3, dil=1, off=9.0),    # wide map, most samples leave the staged patch
    dict(Ci=16, Co=12, H=9, W=9, k=3, dil=4, off=1.0),      # Co = 12, tiny map
    dict(Ci=16, Co=16, H=12, W=12, k=3, dil=1, off=2.0),    # Co > 12: columns + GEMM fallback inside the entry point
    dict(Ci=8, Co=9, H=12, W=12, k=1, dil=1, off=2.0),      # 1x1: fallback
], ids=lambda d: "c%d_co%d_%dx%d_k%d" % (d["Ci"], d["Co"], d["H"], d["W"], d["k"]))
def test_direct_forward_with_sigmoid_matches_composition(geo):
    """ddcn_deform_conv_direct_forward_f32 (no counterpart in the reference: PY:457-458 fused) against
    sigmoid(deform_conv(...)) of the ordinary entry points, forward and all gradients, and against the oracle."""
    mod = importlib.import_module("semantic-segmentation_amd.dcn.deform_conv")
    c = cases.C("direct", N=2, Ci=geo["Ci"], Co=geo["Co"], H=geo["H"], W=geo["W"], kh=geo["k"], kw=geo["k"],
                pad=geo["dil"] * (geo["k"] // 2), dil=geo["dil"], bias=True, modulated=False, off_scale=geo["off"], seed=77)
    inp = _r32(cases.make_inputs(c))
    kw = cases.kwargs(c)
    pre = orc.deform_forward(inp["x"], inp["offset"], None, inp["weight"], inp["bias"], **kw)
    exp = 1.0 / (1.0 + np.exp(-pre))
    res = []
    for fused in (True, False):
        x, off, w, b = _t(inp["x"], True), _t(inp["offset"], True), _t(inp["weight"], True), _t(inp["bias"], True)
        if fused:
            out = mod._DeformConvSigmoidFunction.apply(x, off, w, b, 1, c["pad"], c["dil"], 1, 1)
        else:
            out = torch.sigmoid(pkg.deform_conv(x, off, w, b, 1, c["pad"], c["dil"], 1, 1))
        out.backward(_t(inp["gout"]))
        res.append([t.detach().cpu().numpy() for t in (out, x.grad, off.grad, w.grad, b.grad)])
    assert cases.relerr(res[0][0], exp) <= FWD_TOL
    for got, ref in zip(res[0], res[1]):
        assert cases.relerr(got, ref.astype(np.float64)) <= BWD_TOL


_FOOTPRINTS = {"fx": {}, "xq": {"DDCN_SCATTER_FX": "0"}, "f64": {"DDCN_SCATTER_FX": "0", "DDCN_SCATTER_XQ": "0"}}


@pytest.mark.parametrize("mode", sorted(_FOOTPRINTS))
@pytest.mark.parametrize("c", [cases.MEDIUM_CASES[0], cases.MEDIUM_CASES[1]], ids=lambda c: c["name"])
def test_scatter_footprint_formats(c, mode, monkeypatch):
    """The LDS footprint of the backward scatter in its three formats: packed 32-bit fixed point with per-workgroup
    headroom (the default: two channels per ds_add_u64), exact 64-bit fixed point (DDCN_SCATTER_FX=0) and doubles
    (... DDCN_SCATTER_XQ=0).  Same bar for all; a NaN in grad_output must still surface in grad_input."""
    for k, v in _FOOTPRINTS[mode].items():
        monkeypatch.setenv(k, v)
    inp = cases.make_inputs(c)
    kw = cases.kwargs(c)
    r = _r32(inp)
    exp = orc.deform_backward(r["x"], r["offset"], r["mask"], r["weight"], r["gout"], c["bias"], **kw)
    _, grads = _run_hip(c, inp)
    for k, e in zip(("gx", "goff", "gmask", "gw"), exp[:4]):
        if e is not None:
            assert cases.relerr(grads[k], e) <= BWD_TOL, (k, cases.relerr(grads[k], e))
    bad = dict(inp, gout=inp["gout"].copy())
    bad["gout"][0, 0, 2, 3] = np.nan
    _, grads = _run_hip(c, bad)
    assert np.isnan(grads["gx"]).any()


@pytest.mark.parametrize("scale", [1e-6, 1.0, 1e6], ids=["gout_1e-6", "gout_1", "gout_1e6"])
def test_scatter_fixed_point_is_scale_free_and_as_accurate_as_fp32_atomics(scale):
    """The packed fixed-point footprint scales every chunk by its own largest value (a power of two) and sizes its headroom
    from the workgroup's own samples: grad_input must meet the bar for grad_output of any magnitude, and -- the point of the
    per-workgroup headroom -- sit at fp32-rounding level, not at the 1e-5 of a worst-case (2304 samples per cell) scale."""
    c = cases.C("fx_scale", N=2, Ci=32, Co=24, H=37, W=41, kh=3, kw=3, pad=2, dil=2, modulated=True, off_scale=2.0, seed=5)
    inp = cases.make_inputs(c)
    inp["gout"] = inp["gout"] * scale
    r = _r32(inp)
    exp = orc.deform_backward(r["x"], r["offset"], r["mask"], r["weight"], r["gout"], c["bias"], **cases.kwargs(c))
    _, grads = _run_hip(c, inp)
    for k, e in zip(("gx", "goff", "gmask", "gw"), exp[:4]):
        assert cases.relerr(grads[k], e) <= BWD_TOL, (k, cases.relerr(grads[k], e))
    assert cases.relerr(grads["gx"], exp[0]) <= 2e-6, cases.relerr(grads["gx"], exp[0])


def test_scatter_fixed_point_headroom_when_every_sample_hits_one_cell():
    """The adversarial offset field for a packed footprint: every sample of every pixel lands in the SAME bilinear cell, so
    the tiles around it add 256 pixels x 9 taps into one LDS word per channel pair.  The headroom count sees that (2304
    samples -> 12 bits) and the 32-bit fields cannot wrap: grad_input still meets the bar."""
    c = cases.C("fx_pile", N=1, Ci=8, Co=8, H=24, W=24, kh=3, kw=3, pad=1, dil=1, modulated=True, off_scale=0.0, seed=9)
    inp = cases.make_inputs(c)
    Ho, Wo, K2 = cases.geometry(c)
    off = np.zeros((1, 2 * K2, Ho, Wo), dtype=np.float64)
    ty, tx = 11.25 + 2.0 ** -11, 12.5 + 2.0 ** -11
    for k in range(K2):
        i, j = divmod(k, 3)
        hh = np.arange(Ho)[:, None] - 1 + i
        ww = np.arange(Wo)[None, :] - 1 + j
        off[0, 2 * k] = ty - hh
        off[0, 2 * k + 1] = tx - ww
    inp["offset"] = off
    inp["gout"] = np.abs(inp["gout"]) + 0.5          # same sign: the sums really grow to ~2304 x the largest value
    r = _r32(inp)
    exp = orc.deform_backward(r["x"], r["offset"], r["mask"], r["weight"], r["gout"], c["bias"], **cases.kwargs(c))
    _, grads = _run_hip(c, inp)
    for k, e in zip(("gx", "goff", "gmask", "gw"), exp[:4]):
        assert cases.relerr(grads[k], e) <= BWD_TOL, (k, cases.relerr(grads[k], e))


def _fuzz_cases():
    """Seeded random geometries that steer the launch planners through their branches: strip / 2-D tile im2col, tile9 /
    generic scatter, wide / narrow / idle GEMM tiles, strides, odd sizes, wide and tall maps, large offsets (escapees)."""
    rng = np.random.default_rng(2024)
    out = []
    for i in range(14):
        k = 3 if i % 7 else (1 if i % 2 else 5)          # mostly 3x3, one 1x1, one 5x5
        dil = int(rng.integers(1, 4))
        stride = int(rng.choice([1, 1, 2]))
        groups = int(rng.choice([1, 1, 2]))
        dg = int(rng.choice([1, 2, 4]))
        Ci = int(rng.choice([8, 16, 24, 40])) // (groups * dg) * (groups * dg) or groups * dg * 2
        Ci = max(Ci, groups * dg * 2)
        Co = int(rng.choice([6, 10, 16])) * groups
        H = int(rng.integers(9, 70))
        W = int(rng.choice([int(rng.integers(9, 60)), int(rng.integers(120, 210))]))
        out.append(cases.C("fuzz%02d_k%d_c%d_%dx%d_s%d_d%d_g%d_dg%d" % (i, k, Ci, H, W, stride, dil, groups, dg),
                           N=int(rng.integers(1, 3)), Ci=Ci, Co=Co, H=H, W=W, kh=k, kw=k, stride=stride,
                           pad=dil * (k // 2), dil=dil, groups=groups, dg=dg, bias=bool(i % 2),
                           modulated=bool(i % 3), off_scale=float(rng.choice([0.0, 1.0, 2.0, 9.0])), seed=100 + i))
    return out


@pytest.mark.parametrize("c", _fuzz_cases(), ids=lambda c: c["name"])
def test_fuzz_geometries_against_oracle(c):
    inp = _r32(cases.make_inputs(c))
    kw = cases.kwargs(c)
    exp = dict(out=orc.deform_forward(inp["x"], inp["offset"], inp["mask"], inp["weight"], inp["bias"], **kw))
    exp["gx"], exp["goff"], exp["gmask"], exp["gw"], exp["gb"] = orc.deform_backward(
        inp["x"], inp["offset"], inp["mask"], inp["weight"], inp["gout"], c["bias"], **kw)
    out, grads = _run_hip(c, inp)
    _assert_case(c, out, grads, exp)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.5e-2)], ids=["f16", "bf16"])
def test_half_and_bfloat16_tensors_are_computed_in_fp32(dtype, tol):
    """The reference dispatches FLOATING_TYPES_AND_HALF (deform_conv_cuda_kernel.cu:258); here low-precision tensors
    are converted, computed by the fp32 kernels and converted back.  Expected value: the fp64 oracle on the SAME
    low-precision-rounded inputs; the tolerance is the output rounding of the dtype (parity unpinned by the
    reference, which has no half-precision test)."""
    c = cases.MEDIUM_CASES[0]
    inp = cases.make_inputs(c)
    rnd = {k: (None if v is None else torch.from_numpy(v).to(dtype).to(torch.float64).numpy()) for k, v in inp.items()}
    kw = cases.kwargs(c)
    eout = orc.deform_forward(rnd["x"], rnd["offset"], rnd["mask"], rnd["weight"], rnd["bias"], **kw)
    egx, egoff, egmask, egw, _ = orc.deform_backward(rnd["x"], rnd["offset"], rnd["mask"], rnd["weight"], rnd["gout"],
                                                     c["bias"], **kw)
    t = lambda a: torch.from_numpy(a).to(DEV).to(dtype).requires_grad_(True)
    x, off, m, w = t(inp["x"]), t(inp["offset"]), t(inp["mask"]), t(inp["weight"])
    out = pkg.modulated_deform_conv(x, off, m, w, None, c["stride"], c["pad"], c["dil"], c["groups"], c["dg"])
    assert out.dtype == dtype
    out.backward(torch.from_numpy(inp["gout"]).to(DEV).to(dtype))
    assert cases.relerr(out.detach().float().cpu().numpy(), eout) <= tol
    for got, exp in ((x.grad, egx), (off.grad, egoff), (m.grad, egmask), (w.grad, egw)):
        assert got.dtype == dtype
        assert cases.relerr(got.float().cpu().numpy(), exp) <= tol
    # double tensors have their own kernels (tests/test_gpu_f64.py); mixing dtypes in one call is refused
    assert pkg.modulated_deform_conv(x.double(), off.double(), m.double(), w.double(), None, 1, c["pad"], c["dil"], 1,
                                     1).dtype == torch.float64
    with pytest.raises(RuntimeError):
        pkg.modulated_deform_conv(x.float(), off, m, w, None, 1, c["pad"], c["dil"], 1, 1)   # mixed dtypes


# ---------------------------------------------------------------- the reference's own self-tests, on the GPU
def test_ref_selftest_zero_offset_equals_conv2d_gpu():
    # detseg/ops/dcn/test.py:17-94 (a),(b)
    torch.manual_seed(3)
    pcn = torch.nn.Conv2d(4, 4, 3, stride=1, padding=1, dilation=1, groups=2).to(DEV)
    for mod in (False, True):
        cls = pkg.ModulatedDeformConv if mod else pkg.DeformConv
        dcn = cls(4, 4, (3, 3), stride=1, padding=1, dilation=1, groups=2, deformable_groups=1, bias=True).to(DEV)
        dcn.weight, dcn.bias = pcn.weight, pcn.bias
        x = torch.randn(2, 4, 4, 4, device=DEV)
        offset = torch.zeros(2, 18, 4, 4, device=DEV)
        if mod:
            out = dcn(x, offset, torch.sigmoid(torch.zeros(2, 9, 4, 4, device=DEV)) * 2)
        else:
            out = dcn(x, offset)
        assert (out - pcn(x)).abs().max().item() < 1e-5


def test_ref_selftest_identity_weights_gpu():
    # detseg/ops/dcn/test.py:112-181 (c),(d)
    torch.manual_seed(3)
    groups = 2
    for mod in (False, True):
        cls = pkg.ModulatedDeformConv if mod else pkg.DeformConv
        dcn = cls(4, 4, (3, 3), stride=1, padding=1, dilation=1, groups=groups, deformable_groups=1, bias=True).to(DEV)
        with torch.no_grad():
            dcn.weight.zero_()
            dcn.bias.zero_()
            oc = 4 // groups
            for p in range(dcn.weight.shape[1]):
                for q in range(4):
                    if p == q % oc:
                        dcn.weight[q, p, 1, 1] = 1.0
        x = torch.randn(2, 4, 4, 4, device=DEV)
        offset = torch.zeros(2, 18, 4, 4, device=DEV)
        if mod:
            out = dcn(x, offset, torch.sigmoid(torch.zeros(2, 9, 4, 4, device=DEV))) * 2
        else:
            out = dcn(x, offset)
        assert (x - out).abs().max().item() < 1e-10


# ---------------------------------------------------------------- DDCN layer and packs
def test_ddcn_layer_forward_backward_matches_composition():
    """DepthDeformConvPack == conv_offset -> DeformConv mask branch -> modulated op (deform_conv.py:457-463),
    checked against the oracle evaluated on the layer's own offset / mask tensors."""
    torch.manual_seed(0)
    layer = pkg.DepthDeformConvPack(16, 16, 3, stride=1, padding=2, dilation=2, deformable_groups=1, bias=False,
                                    depth_channels=8).to(DEV)
    with torch.no_grad():   # non-trivial offsets and masks
        layer.conv_offset.weight.normal_(0, 0.05)
        layer.conv_offset.bias.normal_(0, 0.5)
        layer.conv_mask.weight.normal_(0, 0.1)
        layer.conv_mask.bias.normal_(0, 0.1)
    x = torch.randn(2, 16, 19, 17, device=DEV, requires_grad=True)
    depth = torch.randn(2, 8, 19, 17, device=DEV, requires_grad=True)
    out = layer(x, depth)
    out.sum().backward()
    assert out.shape == (2, 16, 19, 17)
    assert layer.offset.shape == (2, 18, 19, 17) and layer.mask.shape == (2, 9, 19, 17)
    f64 = lambda t: t.detach().cpu().numpy().astype(np.float64)
    # mask branch
    emask_pre = orc.deform_forward(f64(depth), f64(layer.offset), None, f64(layer.conv_mask.weight),
                                   f64(layer.conv_mask.bias), 1, 2, 2, 1, 1)
    emask = 1.0 / (1.0 + np.exp(-emask_pre))
    assert cases.relerr(f64(layer.mask), emask) <= FWD_TOL
    eout = orc.deform_forward(f64(x), f64(layer.offset), f64(layer.mask), f64(layer.weight), None, 1, 2, 2, 1, 1)
    assert cases.relerr(f64(out), eout) <= FWD_TOL
    for p in (layer.weight, layer.conv_offset.weight, layer.conv_mask.weight, layer.conv_mask.bias):
        assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().max() > 0
    assert torch.isfinite(x.grad).all() and torch.isfinite(depth.grad).all()


def test_modulated_pack_dg4_dilated_and_deform_pack():
    torch.manual_seed(1)
    mp = pkg.ModulatedDeformConvPack(16, 8, 3, stride=1, padding=2, dilation=2, deformable_groups=4, bias=False).to(DEV)
    with torch.no_grad():
        mp.conv_offset_mask.weight.normal_(0, 0.05)
    x = torch.randn(1, 16, 13, 11, device=DEV, requires_grad=True)
    out = mp(x)
    out.square().sum().backward()
    f64 = lambda t: t.detach().cpu().numpy().astype(np.float64)
    eout = orc.deform_forward(f64(x), f64(mp.offset), f64(mp.mask), f64(mp.weight), None, 1, 2, 2, 1, 4)
    assert cases.relerr(f64(out), eout) <= FWD_TOL
    dp = pkg.DeformConvPack(16, 8, 3, stride=2, padding=1, bias=True).to(DEV)   # C1: runs at all
    y = dp(x.detach())
    assert y.shape == (1, 8, 7, 6)
    ref = torch.nn.functional.conv2d(x.detach(), dp.weight, dp.bias, stride=2, padding=1)   # zero-init offsets
    assert (y - ref).abs().max().item() < 1e-4


def test_error_behaviour_gpu():
    x = torch.randn(2, 4, 5, 5, device=DEV)
    w = torch.randn(4, 4, 3, 3, device=DEV)
    with pytest.raises(RuntimeError, match="invalid shape of offset"):
        pkg.deform_conv(x, torch.zeros(2, 18, 4, 4, device=DEV), w, None, 1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError, match="contiguous"):       # deform_conv_cuda.cpp:179
        pkg.deform_conv(x.transpose(2, 3), torch.zeros(2, 18, 5, 5, device=DEV), w, None, 1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError, match="wont match"):       # deform_conv_cuda.cpp:196-198
        pkg.deform_conv(x, torch.zeros(2, 18, 5, 5, device=DEV), w, None, 1, 1, 1, 2, 1)
    with pytest.raises(RuntimeError, match="float32"):          # an integer tensor names the dtypes that are served
        pkg.deform_conv(x.int(), torch.zeros(2, 18, 5, 5, device=DEV).int(), w.int(), None, 1, 1, 1, 1, 1)
    L = _lib.lib()
    g = _lib.Geom(2, 4, 5, 5, 4, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 5, 5)
    ws = torch.empty(16, device=DEV)
    rc = L.ddcn_deform_conv_forward_f32(_lib.ptr(x), _lib.ptr(w), None, _lib.ptr(x), _lib.ptr(x), _lib.ptr(ws), 64, g,
                                        _lib.current_stream(torch.device(DEV)))
    assert rc == -3 and b"workspace too small" in L.ddcn_last_error()
    rc = L.ddcn_im2col_f32(None, _lib.ptr(x), None, _lib.ptr(x), g, None)
    assert rc == -1


def test_non_default_stream_and_no_grad_inputs():
    c = cases.SMALL_CASES[3]
    g = load_golden(c["name"])
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        out, grads = _run_hip(c, g)
    _assert_case(c, out, grads, g)
    # forward only, nothing requires grad: nothing saved, no backward
    out2 = pkg.modulated_deform_conv(_t(g["x"]), _t(g["offset"]), _t(g["mask"]), _t(g["weight"]), _t(g["bias"]),
                                     c["stride"], c["pad"], c["dil"], c["groups"], c["dg"])
    assert not out2.requires_grad
    assert cases.relerr(out2.cpu().numpy(), g["out"]) <= FWD_TOL


# ---------------------------------------------------------------- full size (BASELINE config 1 layer)
def _config1(n=2, seed=0, off_scale=2.0):
    gen = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(n, 512, 97, 97, generator=gen)
    w = (torch.rand(512, 512, 3, 3, generator=gen) * 2 - 1) / (4608 ** 0.5)
    # odd multiples of 2^-11: tap position + offset is exact in fp32 and never an integer (see cases.make_inputs)
    off = (torch.floor(torch.randn(n, 18, 97, 97, generator=gen) * off_scale * 1024) + 0.5) / 1024
    mask = torch.sigmoid(torch.randn(n, 9, 97, 97, generator=gen))
    gout = torch.randn(n, 512, 97, 97, generator=gen)
    return x, w, off, mask, gout


def test_full_size_config1_against_oracle_one_image():
    """One image of the res5 layer (C=512->512, 97x97, dil=pad=4) against the fp32 oracle run on the host
    (the fp64 oracle takes the same few seconds; fp64 is used so the bar is the stated one)."""
    x, w, off, mask, gout = _config1(n=1, seed=1)
    xd, wd, od, md = (t.to(DEV).requires_grad_(True) for t in (x, w, off, mask))
    out = pkg.modulated_deform_conv(xd, od, md, wd, None, 1, 4, 4, 1, 1)
    out.backward(gout.to(DEV))
    f64 = lambda t: t.numpy().astype(np.float64)
    eout = orc.deform_forward(f64(x), f64(off), f64(mask), f64(w), None, 1, 4, 4, 1, 1)
    egx, egoff, egmask, egw, _ = orc.deform_backward(f64(x), f64(off), f64(mask), f64(w), f64(gout), False, 1, 4, 4, 1, 1)
    assert cases.relerr(out.detach().cpu().numpy(), eout) <= FWD_TOL
    assert cases.relerr(xd.grad.cpu().numpy(), egx) <= BWD_TOL
    assert cases.relerr(od.grad.cpu().numpy(), egoff) <= BWD_TOL
    assert cases.relerr(md.grad.cpu().numpy(), egmask) <= BWD_TOL
    assert cases.relerr(wd.grad.cpu().numpy(), egw) <= BWD_TOL


@pytest.mark.parametrize("name,C,H,stride,dil", [
    ("r101_layer2_block0", 128, 193, 2, 1),    # the strided DDCN layer: 193^2 -> 97^2
    ("r101_layer3_block1", 256, 97, 1, 2),     # the shape 23 of the 30 layers of DDCN-R101 run at
], ids=lambda v: v if isinstance(v, str) else None)
def test_full_size_model_layer_shapes_against_oracle(name, C, H, stride, dil):
    """One image of the DDCN-ResNet101 layer shapes that differ from the config 1 layer, against the fp64 oracle."""
    gen = torch.Generator(device="cpu").manual_seed(7)
    Ho = (H + 2 * dil - (2 * dil + 1)) // stride + 1
    x = torch.randn(1, C, H, H, generator=gen)
    w = (torch.rand(C, C, 3, 3, generator=gen) * 2 - 1) / ((C * 9) ** 0.5)
    off = (torch.floor(torch.randn(1, 18, Ho, Ho, generator=gen) * 2 * 1024) + 0.5) / 1024
    mask = torch.sigmoid(torch.randn(1, 9, Ho, Ho, generator=gen))
    gout = torch.randn(1, C, Ho, Ho, generator=gen)
    xd, wd, od, md = (t.to(DEV).requires_grad_(True) for t in (x, w, off, mask))
    out = pkg.modulated_deform_conv(xd, od, md, wd, None, stride, dil, dil, 1, 1)
    assert out.shape == (1, C, Ho, Ho)
    out.backward(gout.to(DEV))
    f64 = lambda t: t.numpy().astype(np.float64)
    eout = orc.deform_forward(f64(x), f64(off), f64(mask), f64(w), None, stride, dil, dil, 1, 1)
    egx, egoff, egmask, egw, _ = orc.deform_backward(f64(x), f64(off), f64(mask), f64(w), f64(gout), False, stride, dil,
                                                     dil, 1, 1)
    assert cases.relerr(out.detach().cpu().numpy(), eout) <= FWD_TOL
    for got, exp in ((xd.grad, egx), (od.grad, egoff), (md.grad, egmask), (wd.grad, egw)):
        assert cases.relerr(got.cpu().numpy(), exp) <= BWD_TOL


def test_full_size_config1_properties():
    """Size-independent properties at BASELINE config 1 (N=2): zero offsets + unit mask == dilated conv2d;
    linearity in the weight; gradient of sum(out) w.r.t. bias-free weight equals the summed columns identity
    <gW, W> == <out, gout> (Euler identity for a map linear in W)."""
    x, w, off, mask, gout = _config1(n=2, seed=2)
    xd, wd, od, md, gd = (t.to(DEV) for t in (x, w, off, mask, gout))
    zero = torch.zeros_like(od)
    one = torch.ones_like(md)
    out0 = pkg.modulated_deform_conv(xd, zero, one, wd, None, 1, 4, 4, 1, 1)
    ref = torch.nn.functional.conv2d(xd.double(), wd.double(), None, 1, 4, 4).float()
    assert ((out0 - ref).abs().max() / ref.abs().max()).item() <= FWD_TOL
    # linearity in W
    w2 = torch.randn_like(wd) / 68.0
    o1 = pkg.modulated_deform_conv(xd, od, md, wd, None, 1, 4, 4, 1, 1)
    o2 = pkg.modulated_deform_conv(xd, od, md, w2, None, 1, 4, 4, 1, 1)
    o12 = pkg.modulated_deform_conv(xd, od, md, wd + 0.5 * w2, None, 1, 4, 4, 1, 1)
    assert ((o12 - (o1 + 0.5 * o2)).abs().max() / o12.abs().max()).item() <= 1e-5
    # <gW, W> == <out, gout>  and  <gx, x> == <out, gout>  (the op is linear in W and in x)
    wg = wd.clone().requires_grad_(True)
    xg = xd.clone().requires_grad_(True)
    o = pkg.modulated_deform_conv(xg, od, md, wg, None, 1, 4, 4, 1, 1)
    o.backward(gd)
    lhs = (o.detach().double() * gd.double()).sum()
    assert abs(((wg.grad.double() * wd.double()).sum() - lhs) / lhs).item() <= 1e-4
    assert abs(((xg.grad.double() * xd.double()).sum() - lhs) / lhs).item() <= 1e-4


# ---------------------------------------------------------------- persistent (stream-K) fp32 GEMMs
@pytest.mark.parametrize("mode", ["1", "0"], ids=["persistent", "flat_grid"])
@pytest.mark.parametrize("geo", [
    dict(N=3, Ci=64, Co=160, H=40, W=45, groups=1),      # fewer tiles than wave groups: everything is tail
    dict(N=1, Ci=128, Co=136, H=97, W=97, groups=1),     # G2: one whole round + a tail; M, N and K edges
    dict(N=2, Ci=96, Co=320, H=33, W=61, groups=2),      # weight groups; G3 folds both images into K
    dict(N=5, Ci=32, Co=256, H=50, W=50, groups=1),      # five images folded into K
], ids=lambda d: "n%d_c%d_co%d_%dx%d_g%d" % (d["N"], d["Ci"], d["Co"], d["H"], d["W"], d["groups"]))
def test_persistent_gemms_against_fp64_matmul(geo, mode, monkeypatch):
    """G2 and G3 through the persistent kernel (gemm_f32_sk: whole rounds + stream-K tail, images folded into K for G3,
    DDCN_GEMM_SK=1 forces it) and through the flat grid (=0), against fp64 products; G2 twice: its plain stores make
    it bit-reproducible.  Reference call sites: deform_conv_cuda.cpp:503-506 (G2), :539-544 (G3)."""
    monkeypatch.setenv("DDCN_GEMM_SK", mode)
    c = cases.C("sk", N=geo["N"], Ci=geo["Ci"], Co=geo["Co"], H=geo["H"], W=geo["W"], pad=1, dil=1, groups=geo["groups"])
    g = _geom(c)
    L = _lib.lib()
    G, P = geo["groups"], g.Ho * g.Wo
    Kg, Cog = c["Ci"] // G * 9, c["Co"] // G
    ld = L.ddcn_col_ld(g)
    rng = np.random.default_rng(6)
    w = rng.standard_normal((G, Cog, Kg)).astype(np.float32)
    col = rng.standard_normal((c["N"], G, Kg, P)).astype(np.float32)
    gout = rng.standard_normal((c["N"], G, Cog, P)).astype(np.float32)
    col_d = torch.full((c["N"], G * Kg, ld), float("nan"), device=DEV)
    col_d[:, :, :P] = torch.from_numpy(col.reshape(c["N"], G * Kg, P)).to(DEV)
    w_d, gout_d = _t(w), _t(gout)
    s = _lib.current_stream(torch.device(DEV))
    exp = np.einsum("gok,ngop->ngkp", w.astype(np.float64), gout.astype(np.float64)).reshape(c["N"], G * Kg, P)
    runs = []
    for _ in range(2):
        gcol = torch.full((c["N"], G * Kg, ld), float("nan"), device=DEV)
        _lib.check(L.ddcn_gemm_gcol_f32(_lib.ptr(w_d), _lib.ptr(gout_d), _lib.ptr(gcol), g, s))
        runs.append(gcol[:, :, :P].cpu().numpy())
    assert cases.relerr(runs[0], exp) <= FWD_TOL
    assert np.array_equal(runs[0], runs[1])
    gw = torch.ones((G, Cog, Kg), device=DEV)     # accumulate semantics: starts at 1
    _lib.check(L.ddcn_gemm_gweight_f32(_lib.ptr(gout_d), _lib.ptr(col_d), _lib.ptr(gw), g, s))
    exp = 1.0 + np.einsum("ngop,ngkp->gok", gout.astype(np.float64), col.astype(np.float64))
    assert cases.relerr(gw.cpu().numpy(), exp) <= BWD_TOL
