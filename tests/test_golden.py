"""CPU: the oracle reproduces the committed golden fixtures (fp64 exactly-ish, fp32 within the parity bar)."""
import numpy as np
import pytest

import cases
from helpers import BWD_TOL, FWD_TOL, load_golden
from oracle import ddcn_oracle as orc


@pytest.mark.parametrize("c", cases.SMALL_CASES, ids=lambda c: c["name"])
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_oracle_reproduces_golden(c, dt):
    g = load_golden(c["name"])
    f = lambda k: g[k].astype(dt) if k in g else None
    kw = cases.kwargs(c)
    out = orc.deform_forward(f("x"), f("offset"), f("mask"), f("weight"), f("bias"), **kw)
    gx, goff, gmask, gw, gb = orc.deform_backward(f("x"), f("offset"), f("mask"), f("weight"), f("gout"),
                                                  c["bias"], **kw)
    ft, bt = (1e-13, 1e-12) if dt == np.float64 else (FWD_TOL, BWD_TOL)
    assert cases.relerr(out, g["out"]) < ft
    assert cases.relerr(gx, g["gx"]) < bt
    assert cases.relerr(goff, g["goff"]) < bt
    assert cases.relerr(gw, g["gw"]) < bt
    if c["modulated"]:
        assert cases.relerr(gmask, g["gmask"]) < bt
    if c["bias"]:
        assert cases.relerr(gb, g["gb"]) < bt


@pytest.mark.parametrize("c", cases.DEPTHAWARE_CASES, ids=lambda c: c["name"])
def test_oracle_reproduces_golden_depthaware(c):
    g = load_golden(c["name"])
    out = orc.depthaware_forward(g["x"], g["depth"], g["weight"], g.get("bias"), c["stride"], c["pad"], c["dil"],
                                 c["groups"])
    gx, gw, gb = orc.depthaware_backward(g["x"], g["depth"], g["weight"], g["gout"], c["bias"], c["stride"],
                                         c["pad"], c["dil"], c["groups"])
    assert cases.relerr(out, g["out"]) < 1e-13
    assert cases.relerr(gx, g["gx"]) < 1e-12
    assert cases.relerr(gw, g["gw"]) < 1e-12
    if c["bias"]:
        assert cases.relerr(gb, g["gb"]) < 1e-12
