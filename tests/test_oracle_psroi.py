"""Pins for the deformable PS-RoI pooling restatement (oracle/psroi_ref_impl.inc).

The reference has no test, fixture or caller for this operator (detseg/ops/dcn/deform_pool.py is only
re-exported by detseg/ops/dcn/__init__.py): PARITY UNPINNED by reference data.  The pins here are
(1) an independently written grid_sample/autograd formulation (tests/indep_ref.py), forward and both
gradients, in fp64, and (2) central finite differences of the oracle's own fp64 forward -- the strategy
the reference's detseg/ops/dcn/test.py uses for the convolution (gradcheck).
"""
import numpy as np
import pytest
import torch

import indep_ref
from oracle import ddcn_oracle as orc
from psroi_cases import PSROI_CASES, make_psroi_inputs


@pytest.mark.parametrize("case", PSROI_CASES, ids=lambda c: c["name"])
def test_forward_and_gradients_match_independent_formulation(case):
    data, rois, offset, gout, kw = make_psroi_inputs(case, np.float64)
    out, cnt = orc.psroi_forward(data, rois, offset, **kw)
    gdata, goff = orc.psroi_backward(gout, data, rois, offset, cnt, **kw)

    td = torch.from_numpy(data).requires_grad_(True)
    to = torch.from_numpy(offset).requires_grad_(not kw["no_trans"])
    ref, ref_cnt = indep_ref.deform_psroi_pool(td, torch.from_numpy(rois), to, **kw)
    np.testing.assert_array_equal(cnt, ref_cnt.numpy())
    np.testing.assert_allclose(out, ref.detach().numpy(), rtol=1e-10, atol=1e-12)
    if ref.numel() == 0:
        return
    ref.backward(torch.from_numpy(gout))
    np.testing.assert_allclose(gdata, td.grad.numpy(), rtol=1e-9, atol=1e-11)
    if not kw["no_trans"]:
        np.testing.assert_allclose(goff, to.grad.numpy(), rtol=1e-8, atol=1e-10)


def test_offset_gradient_matches_finite_differences():
    case = dict(name="fd", B=2, out_channels=4, group_size=2, H=11, W=13, R=3, out_size=3, part_size=3,
                sample_per_part=2, trans_std=0.2, spatial_scale=0.3, classes=2, no_trans=False, seed=5)
    data, rois, offset, gout, kw = make_psroi_inputs(case, np.float64)
    out, cnt = orc.psroi_forward(data, rois, offset, **kw)
    _, goff = orc.psroi_backward(gout, data, rois, offset, cnt, **kw)
    eps = 1e-6
    fd = np.zeros_like(offset)
    for i in range(offset.size):
        d = np.zeros(offset.size)
        d[i] = eps
        d = d.reshape(offset.shape)
        hi, _ = orc.psroi_forward(data, rois, offset + d, **kw)
        lo, _ = orc.psroi_forward(data, rois, offset - d, **kw)
        fd.flat[i] = ((hi - lo) * gout).sum() / (2 * eps)
    np.testing.assert_allclose(goff, fd, rtol=1e-5, atol=1e-7)


def test_data_gradient_is_the_transpose_of_the_forward():
    # forward is linear in data: <gout, F(data)> == <F^T(gout), data>
    case = PSROI_CASES[0]
    data, rois, offset, gout, kw = make_psroi_inputs(case, np.float64)
    out, cnt = orc.psroi_forward(data, rois, offset, **kw)
    gdata, _ = orc.psroi_backward(gout, data, rois, offset, cnt, **kw)
    assert abs((gout * out).sum() - (gdata * data).sum()) <= 1e-10 * abs((gout * out).sum())


def test_constant_map_pools_to_the_constant_and_counts_samples():
    data = np.full((1, 4, 9, 9), 3.25, dtype=np.float32)
    rois = np.array([[0, 8, 8, 40, 40], [0, -400, -400, -300, -300]], dtype=np.float32)
    out, cnt = orc.psroi_forward(data, rois, np.zeros(0, np.float32), 0.125, 2, 1, True, group_size=2,
                                 sample_per_part=3)
    np.testing.assert_array_equal(cnt[0], 9)
    np.testing.assert_allclose(out[0], 3.25, rtol=1e-6)
    # a RoI entirely off the map: every sample dropped, output 0, count 0 (deform_pool_cuda_kernel.cu:136)
    np.testing.assert_array_equal(cnt[1], 0)
    np.testing.assert_array_equal(out[1], 0)


def test_f32_tracks_f64():
    for case in PSROI_CASES[:3]:
        d64, r64, o64, g64, kw = make_psroi_inputs(case, np.float64)
        out64, cnt64 = orc.psroi_forward(d64, r64, o64, **kw)
        out32, cnt32 = orc.psroi_forward(d64.astype(np.float32), r64.astype(np.float32), o64.astype(np.float32), **kw)
        np.testing.assert_array_equal(cnt32, cnt64)
        np.testing.assert_allclose(out32, out64, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("case", [c for c in PSROI_CASES if c["R"]], ids=lambda c: c["name"])
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_oracle_reproduces_psroi_golden(case, dt):
    from helpers import BWD_TOL, FWD_TOL, load_golden
    import cases as conv_cases
    g = load_golden("psroi_" + case["name"])
    _, _, _, _, kw = make_psroi_inputs(case, np.float32)
    off = g["offset"].astype(dt) if "offset" in g else np.zeros(0, dt)
    out, cnt = orc.psroi_forward(g["data"].astype(dt), g["rois"].astype(dt), off, **kw)
    gdata, goff = orc.psroi_backward(g["gout"].astype(dt), g["data"].astype(dt), g["rois"].astype(dt), off, cnt, **kw)
    ft, bt = (1e-13, 1e-12) if dt == np.float64 else (FWD_TOL, BWD_TOL)
    np.testing.assert_array_equal(cnt, g["count"])
    assert conv_cases.relerr(out, g["out"]) < ft
    assert conv_cases.relerr(gdata, g["gdata"]) < bt
    if goff is not None:
        assert conv_cases.relerr(goff, g["goff"]) < bt
