"""GPU parity of deformable PS-RoI pooling: the HIP kernels (through the C ABI) against the CPU oracle and the
committed fixtures.  Same bar as the convolution: fp32 forward rel-err <= 1e-5, backward <= 1e-4; the per-bin
sample counts are integers and must match exactly.

The oracle behind the expectations is "parity unpinned" by reference data (the reference has no test for this
operator); see tests/test_oracle_psroi.py for what pins it instead.
"""
import importlib

import numpy as np
import pytest
import torch

import cases
from helpers import BWD_TOL, FWD_TOL, load_golden
from oracle import ddcn_oracle as orc
from psroi_cases import PSROI_CASES, make_psroi_inputs

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    pkg = importlib.import_module("semantic-segmentation_amd")
    _lib = importlib.import_module("semantic-segmentation_amd.dcn._lib")
DEV = "cuda:0"


def _t(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
    return t.requires_grad_(True) if grad else t


def _args(kw):
    return (kw["spatial_scale"], kw["out_size"], kw["out_channels"], kw["no_trans"], kw["group_size"],
            kw["part_size"], kw["sample_per_part"], kw["trans_std"])


def _run(data, rois, offset, gout, kw):
    d, o = _t(data, True), _t(offset, not kw["no_trans"])
    out = pkg.deform_roi_pooling(d, _t(rois), o, *_args(kw))
    out.backward(_t(gout))
    torch.cuda.synchronize()
    goff = None if kw["no_trans"] else o.grad.cpu().numpy()
    return out.detach().cpu().numpy(), d.grad.cpu().numpy(), goff


def _count_through_c_abi(data, rois, offset, kw):
    d, r = _t(data), _t(rois)
    o = None if kw["no_trans"] else _t(offset)
    R = rois.shape[0]
    out = torch.empty(R, kw["out_channels"], kw["out_size"], kw["out_size"], device=DEV)
    cnt = torch.empty_like(out)
    g = _lib.RoiGeom(*data.shape, R, kw["out_channels"], kw["out_size"], kw["group_size"], kw["part_size"],
                     kw["sample_per_part"], 2 if o is None else o.shape[1], int(kw["no_trans"]),
                     kw["spatial_scale"], kw["trans_std"])
    _lib.check(_lib.lib().ddcn_deform_psroi_pool_forward_f32(_lib.ptr(d), _lib.ptr(r), _lib.ptr(o), _lib.ptr(out),
                                                             _lib.ptr(cnt), g, _lib.current_stream(d.device)))
    torch.cuda.synchronize()
    return out.cpu().numpy(), cnt.cpu().numpy()


@pytest.mark.parametrize("case", [c for c in PSROI_CASES if c["R"]], ids=lambda c: c["name"])
def test_hip_matches_psroi_golden(case):
    g = load_golden("psroi_" + case["name"])
    _, _, _, _, kw = make_psroi_inputs(case, np.float32)
    off = g["offset"] if "offset" in g else np.zeros(0)
    out, gdata, goff = _run(g["data"], g["rois"], off, g["gout"], kw)
    _, cnt = _count_through_c_abi(g["data"], g["rois"], off, kw)
    np.testing.assert_array_equal(cnt, g["count"])
    assert cases.relerr(out, g["out"]) <= FWD_TOL, cases.relerr(out, g["out"])
    assert cases.relerr(gdata, g["gdata"]) <= BWD_TOL, cases.relerr(gdata, g["gdata"])
    if goff is not None:
        assert cases.relerr(goff, g["goff"]) <= BWD_TOL, cases.relerr(goff, g["goff"])


MEDIUM = [
    # Faster R-CNN style head: 7x7 bins, one offset class, stride-16 map
    dict(name="head_7x7", B=2, out_channels=64, group_size=1, H=38, W=50, R=96, out_size=7, part_size=7,
         sample_per_part=4, trans_std=0.1, spatial_scale=1 / 16, classes=1, no_trans=False, seed=11),
    # R-FCN style: position-sensitive 7x7 groups, 4 offset classes, coarser part grid
    dict(name="rfcn_7x7_g7", B=2, out_channels=8, group_size=7, H=32, W=40, R=64, out_size=7, part_size=4,
         sample_per_part=4, trans_std=0.1, spatial_scale=1 / 16, classes=4, no_trans=False, seed=12),
    # many bins per workgroup pass (bins > 256) and an odd sample count
    dict(name="out17", B=1, out_channels=6, group_size=1, H=40, W=40, R=9, out_size=17, part_size=17,
         sample_per_part=3, trans_std=0.05, spatial_scale=0.125, classes=1, no_trans=False, seed=13),
    dict(name="head_no_trans", B=3, out_channels=48, group_size=1, H=25, W=34, R=40, out_size=7, part_size=7,
         sample_per_part=2, trans_std=0.0, spatial_scale=1 / 8, classes=1, no_trans=True, seed=14),
]


@pytest.mark.parametrize("case", MEDIUM, ids=lambda c: c["name"])
def test_hip_matches_oracle_medium(case):
    data, rois, offset, gout, kw = make_psroi_inputs(case, np.float32)
    out, gdata, goff = _run(data, rois, offset, gout, kw)
    _, cnt = _count_through_c_abi(data, rois, offset, kw)
    f64 = lambda a: a.astype(np.float64)
    eout, ecnt = orc.psroi_forward(f64(data), f64(rois), f64(offset), **kw)
    egdata, egoff = orc.psroi_backward(f64(gout), f64(data), f64(rois), f64(offset), ecnt, **kw)
    # discrete part first: which samples each bin kept (fp32 oracle: same arithmetic as the kernel)
    _, cnt32 = orc.psroi_forward(data, rois, offset, **kw)
    np.testing.assert_array_equal(cnt, cnt32)
    np.testing.assert_array_equal(cnt, ecnt)
    assert cases.relerr(out, eout) <= FWD_TOL, cases.relerr(out, eout)
    assert cases.relerr(gdata, egdata) <= BWD_TOL, cases.relerr(gdata, egdata)
    if goff is not None:
        assert cases.relerr(goff, egoff) <= BWD_TOL, cases.relerr(goff, egoff)


def test_aligned_rois_land_on_exact_quarter_pixels():
    """Power-of-two scale and RoI sizes that make every sample coordinate exactly representable: samples sit ON
    pixel centres and ON the -0.5 / W-0.5 drop boundary, where one ulp decides.  Counts must equal the oracle's."""
    rng = np.random.default_rng(21)
    data = rng.standard_normal((1, 4, 16, 16)).astype(np.float32)
    rois = np.array([[0, 0, 0, 111, 111], [0, 16, 32, 127, 143], [0, 144, 144, 255, 255], [0, 200, 8, 311, 119],
                     [0, 0, 0, 255, 255]], dtype=np.float32)
    offset = np.zeros((5, 2, 7, 7), dtype=np.float32)
    gout = rng.standard_normal((5, 4, 7, 7)).astype(np.float32)
    kw = dict(spatial_scale=1 / 16, out_size=7, out_channels=4, no_trans=False, group_size=1, part_size=7,
              sample_per_part=4, trans_std=0.1)
    out, gdata, goff = _run(data, rois, offset, gout, kw)
    _, cnt = _count_through_c_abi(data, rois, offset, kw)
    eout, ecnt = orc.psroi_forward(data, rois, offset, **kw)
    egdata, egoff = orc.psroi_backward(gout, data, rois, offset, ecnt, **kw)
    np.testing.assert_array_equal(cnt, ecnt)
    assert cnt.min() < 16 and cnt.max() == 16
    assert cases.relerr(out, eout) <= FWD_TOL
    assert cases.relerr(gdata, egdata) <= BWD_TOL
    assert cases.relerr(goff, egoff) <= BWD_TOL


def test_no_rois_and_out_of_batch_roi():
    data = torch.randn(2, 3, 8, 8, device=DEV, requires_grad=True)
    out = pkg.deform_roi_pooling(data, torch.zeros(0, 5, device=DEV), data.new_empty(0), 0.25, 3, 3, True)
    assert out.shape == (0, 3, 3, 3)
    out.sum().backward()
    assert float(data.grad.abs().sum()) == 0
    # batch index 5 of a batch of 2: pools nothing instead of reading out of bounds
    rois = torch.tensor([[5, 0, 0, 20, 20], [1, 0, 0, 20, 20]], dtype=torch.float32, device=DEV)
    data.grad = None
    out = pkg.deform_roi_pooling(data, rois, data.new_empty(0), 0.25, 3, 3, True)
    assert float(out[0].detach().abs().sum()) == 0 and float(out[1].detach().abs().sum()) > 0
    out.sum().backward()
    assert float(data.grad[0].abs().sum()) == 0 and float(data.grad[1].abs().sum()) > 0


def test_error_behaviour_gpu():
    data = torch.randn(1, 8, 8, 8, device=DEV)
    rois = torch.tensor([[0, 0, 0, 20, 20]], dtype=torch.float32, device=DEV)
    empty = data.new_empty(0)
    with pytest.raises(RuntimeError, match="shape mismatch"):       # 8 maps != 3 * 2^2
        pkg.deform_roi_pooling(data, rois, empty, 0.25, 3, 3, True, 2)
    with pytest.raises(RuntimeError, match="rois must be"):
        pkg.deform_roi_pooling(data, rois[:, :4], empty, 0.25, 3, 8, True)
    with pytest.raises(RuntimeError, match="invalid shape of offset"):
        pkg.deform_roi_pooling(data, rois, torch.zeros(1, 2, 4, 4, device=DEV), 0.25, 3, 8, False)
    with pytest.raises(RuntimeError, match="shape mismatch"):       # 3 offset classes do not divide 8 channels
        pkg.deform_roi_pooling(data, rois, torch.zeros(1, 6, 3, 3, device=DEV), 0.25, 3, 8, False)
    with pytest.raises(RuntimeError, match="contiguous"):
        pkg.deform_roi_pooling(data.transpose(2, 3), rois, empty, 0.25, 3, 8, True)
    d = data.clone().requires_grad_(True)
    out = pkg.deform_roi_pooling(d, rois, empty, 0.25, 33, 8, True)  # forward has no size limit ...
    with pytest.raises(RuntimeError, match="unsupported"):            # ... backward keeps bin tables in LDS
        out.sum().backward()


def test_packs_start_from_rigid_pooling_and_train():
    torch.manual_seed(0)
    data = torch.randn(2, 16, 20, 24, device=DEV, requires_grad=True)
    rois = torch.tensor([[0, 10, 12, 200, 150], [1, 40, 8, 300, 260], [1, 0, 0, 380, 310]], dtype=torch.float32,
                        device=DEV)
    rigid = pkg.DeformRoIPooling(1 / 16, 7, 16, True)(data, rois, None)
    m = pkg.DeformRoIPoolingPack(1 / 16, 7, 16, False, trans_std=0.1, deform_fc_channels=64).to(DEV)
    out = m(data, rois)
    # zero-initialised last layer -> zero offsets -> the rigid pooling (deform_pool.py:147-148)
    assert torch.equal(out, rigid)
    mm = pkg.ModulatedDeformRoIPoolingPack(1 / 16, 7, 16, False, trans_std=0.1, deform_fc_channels=64).to(DEV)
    out2 = mm(data, rois)
    assert torch.allclose(out2, rigid * 0.5)   # sigmoid(0) modulation (deform_pool.py:224-225)
    (out.square().sum() + out2.square().sum()).backward()
    assert data.grad is not None and float(data.grad.abs().sum()) > 0
    assert float(m.offset_fc[4].weight.grad.abs().sum()) > 0      # the offset branch receives gradient
    assert float(mm.mask_fc[2].weight.grad.abs().sum()) > 0


@pytest.mark.parametrize("case", [c for c in PSROI_CASES if c["R"]], ids=lambda c: c["name"])
def test_fp64_kernels_match_golden(case):
    g = load_golden("psroi_" + case["name"])
    _, _, _, _, kw = make_psroi_inputs(case, np.float32)
    d = torch.from_numpy(g["data"]).to(DEV).requires_grad_(True)
    o = torch.from_numpy(g["offset"] if "offset" in g else np.zeros(0)).to(DEV).requires_grad_(not kw["no_trans"])
    out = pkg.deform_roi_pooling(d, torch.from_numpy(g["rois"]).to(DEV), o, *_args(kw))
    assert out.dtype == torch.float64
    out.backward(torch.from_numpy(g["gout"]).to(DEV))
    assert cases.relerr(out.detach().cpu().numpy(), g["out"]) <= 1e-12
    assert cases.relerr(d.grad.cpu().numpy(), g["gdata"]) <= 1e-11
    if not kw["no_trans"]:
        assert cases.relerr(o.grad.cpu().numpy(), g["goff"]) <= 1e-11


def test_fp64_gradcheck_of_the_pooling_operator():
    """torch.autograd.gradcheck on the HIP operator in double, with the settings of the reference's convolution
    self-test (detseg/ops/dcn/test.py:208-267) -- the reference has no such test for the pooling."""
    case = dict(name="gc", B=2, out_channels=4, group_size=2, H=9, W=11, R=3, out_size=3, part_size=3,
                sample_per_part=2, trans_std=0.2, spatial_scale=0.3, classes=2, no_trans=False, seed=9)
    data, rois, offset, _, kw = make_psroi_inputs(case, np.float64)
    d = torch.from_numpy(data * 0.1).to(DEV).requires_grad_(True)
    o = torch.from_numpy(offset).to(DEV).requires_grad_(True)
    r = torch.from_numpy(rois).to(DEV)
    fn = lambda a, b: pkg.deform_roi_pooling(a, r, b, *_args(kw))
    assert torch.autograd.gradcheck(fn, (d, o), eps=1e-6, atol=1e-6, rtol=1e-4, raise_exception=True, nondet_tol=1e-9)


def test_half_tensors_are_computed_in_fp32():
    case = MEDIUM[0]
    data, rois, offset, gout, kw = make_psroi_inputs(case, np.float32)
    d = _t(data).half().requires_grad_(True)
    o = _t(offset).half().requires_grad_(True)
    out = pkg.deform_roi_pooling(d, _t(rois).half(), o, *_args(kw))
    assert out.dtype == torch.float16
    out.backward(_t(gout).half())
    assert d.grad.dtype == torch.float16 and o.grad.dtype == torch.float16
    f = lambda t: t.detach().float().cpu().numpy().astype(np.float64)
    eout, ecnt = orc.psroi_forward(f(d), f(_t(rois).half()), f(o), **kw)
    assert cases.relerr(f(out), eout) <= 2e-3


def test_plane_owner_forward_zeroes_out_of_batch_rois_and_matches_direct_kernel(monkeypatch):
    """The LDS plane-owner forward (large launches) against the one-lane-per-bin kernel: same bits; a RoI whose batch
    index names no image is visited by no plane owner and must still come out as zeros."""
    case = MEDIUM[0]
    data, rois, offset, gout, kw = make_psroi_inputs(case, np.float32)
    rois = rois.copy()
    rois[0, 0] = 99
    out_p, cnt_p = _count_through_c_abi(data, rois, offset, kw)
    assert np.all(out_p[0] == 0) and np.all(cnt_p[0] == 0) and np.abs(out_p[1:]).max() > 0
    eout, ecnt = orc.psroi_forward(data[:, :, :, :].astype(np.float64), rois[1:].astype(np.float64),
                                   offset[1:].astype(np.float64), **kw)
    np.testing.assert_array_equal(cnt_p[1:], ecnt)
    assert cases.relerr(out_p[1:], eout) <= FWD_TOL
