"""Generates tests/golden/psroi_*.npz: inputs + expected outputs of deformable PS-RoI pooling.

The reference holds no vectors for this operator (and none can be made from it: CUDA only), so the
expectations are the fp64 CPU oracle's (oracle/psroi_ref_impl.inc), cross-checked here against the independent
grid_sample/autograd formulation (tests/indep_ref.py) before anything is written.  Inputs are stored as fp32;
expectations are fp64 results computed FROM those fp32 values.

Run from the repo root:  python tests/golden/make_golden_psroi.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import indep_ref  # noqa: E402
from oracle import ddcn_oracle as orc  # noqa: E402
from psroi_cases import PSROI_CASES, make_psroi_inputs  # noqa: E402


def main():
    n = 0
    for c in PSROI_CASES:
        if c["R"] == 0:
            continue
        data, rois, offset, gout, kw = (a.astype(np.float64) if isinstance(a, np.ndarray) else a
                                        for a in make_psroi_inputs(c, np.float32))
        out, cnt = orc.psroi_forward(data, rois, offset, **kw)
        gdata, goff = orc.psroi_backward(gout, data, rois, offset, cnt, **kw)
        td = torch.from_numpy(data).requires_grad_(True)
        to = torch.from_numpy(offset).requires_grad_(not kw["no_trans"])
        ref, ref_cnt = indep_ref.deform_psroi_pool(td, torch.from_numpy(rois), to, **kw)
        ref.backward(torch.from_numpy(gout))
        assert np.array_equal(cnt, ref_cnt.numpy()), c["name"]
        assert np.allclose(out, ref.detach().numpy(), rtol=1e-10, atol=1e-12), c["name"]
        assert np.allclose(gdata, td.grad.numpy(), rtol=1e-9, atol=1e-11), c["name"]
        d = dict(data=data.astype(np.float32), rois=rois.astype(np.float32), gout=gout.astype(np.float32), out=out,
                 count=cnt.astype(np.float32), gdata=gdata)
        if not kw["no_trans"]:
            assert np.allclose(goff, to.grad.numpy(), rtol=1e-8, atol=1e-10), c["name"]
            d.update(offset=offset.astype(np.float32), goff=goff)
        np.savez_compressed(os.path.join(HERE, "psroi_" + c["name"] + ".npz"), **d)
        n += 1
    print("wrote", n, "psroi fixtures to", HERE)


if __name__ == "__main__":
    main()
