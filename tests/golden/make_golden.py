"""Generates tests/golden/*.npz: inputs + expected outputs of the DCN ops for the small parity cases.

The reference ships no golden vectors and cannot be built or imported here, so the
expectations come from the fp64 CPU oracle (oracle/), and every file is cross-checked against the
independent grid_sample/autograd formulation (tests/indep_ref.py) before it is written.
Each fixture holds the fp32-rounded inputs (as fp64 arrays) and the fp64 results computed FROM those
rounded inputs, i.e. what an exact fp32-input implementation must reproduce.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import cases  # noqa: E402
import indep_ref  # noqa: E402
from oracle import ddcn_oracle as orc  # noqa: E402


def r32(a):
    return None if a is None else a.astype(np.float32).astype(np.float64)


INPUT_KEYS = ("x", "offset", "mask", "weight", "bias", "gout", "depth")


def save(c, d):
    # inputs are exactly representable in fp32: store them as fp32 (half the bytes), expectations as fp64
    d = {k: (v.astype(np.float32) if k in INPUT_KEYS else v) for k, v in d.items()}
    np.savez_compressed(os.path.join(HERE, c["name"] + ".npz"), **d)


def main():
    for c in cases.SMALL_CASES:
        inp = {k: r32(v) for k, v in cases.make_inputs(c).items()}
        kw = cases.kwargs(c)
        out = orc.deform_forward(inp["x"], inp["offset"], inp["mask"], inp["weight"], inp["bias"], **kw)
        gx, goff, gmask, gw, gb = orc.deform_backward(inp["x"], inp["offset"], inp["mask"], inp["weight"],
                                                      inp["gout"], c["bias"], **kw)
        t = {k: (torch.from_numpy(v).clone().requires_grad_(True) if v is not None else None)
             for k, v in inp.items() if k in ("x", "offset", "mask", "weight", "bias")}
        iout = indep_ref.deform_conv(t["x"], t["offset"], t["mask"], t["weight"], t["bias"], c["stride"], c["pad"],
                                     c["dil"], c["groups"], c["dg"])
        iout.backward(torch.from_numpy(inp["gout"]))
        assert cases.relerr(out, iout.detach().numpy()) < 1e-10, c["name"]
        assert cases.relerr(gx, t["x"].grad.numpy()) < 1e-10, c["name"]
        assert cases.relerr(gw, t["weight"].grad.numpy()) < 1e-10, c["name"]
        if not cases.has_integer_coords(c):
            assert cases.relerr(goff, t["offset"].grad.numpy()) < 1e-10, c["name"]
        d = dict(x=inp["x"], offset=inp["offset"], weight=inp["weight"], gout=inp["gout"], out=out, gx=gx,
                 goff=goff, gw=gw)
        if c["modulated"]:
            d.update(mask=inp["mask"], gmask=gmask)
        if c["bias"]:
            d.update(bias=inp["bias"], gb=gb)
        save(c, d)
    for c in cases.DEPTHAWARE_CASES:
        inp = {k: r32(v) for k, v in cases.make_inputs(c).items()}
        out = orc.depthaware_forward(inp["x"], inp["depth"], inp["weight"], inp["bias"], c["stride"], c["pad"],
                                     c["dil"], c["groups"])
        gx, gw, gb = orc.depthaware_backward(inp["x"], inp["depth"], inp["weight"], inp["gout"], c["bias"],
                                             c["stride"], c["pad"], c["dil"], c["groups"])
        x = torch.from_numpy(inp["x"]).requires_grad_(True)
        iout = indep_ref.depthaware_conv(x, torch.from_numpy(inp["depth"]), torch.from_numpy(inp["weight"]),
                                         None if inp["bias"] is None else torch.from_numpy(inp["bias"]),
                                         c["stride"], c["pad"], c["dil"], c["groups"])
        assert cases.relerr(out, iout.detach().numpy()) < 1e-10, c["name"]
        d = dict(x=inp["x"], depth=inp["depth"], weight=inp["weight"], gout=inp["gout"], out=out, gx=gx, gw=gw)
        if c["bias"]:
            d.update(bias=inp["bias"], gb=gb)
        save(c, d)
    print("wrote", len(cases.SMALL_CASES) + len(cases.DEPTHAWARE_CASES), "fixtures to", HERE)


if __name__ == "__main__":
    main()
