"""Helpers shared by the parity tests."""
import os

import numpy as np

import cases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

FWD_TOL = 1e-5   # BASELINE.json north_star: forward rel-err <= 1e-5 (fp32)
BWD_TOL = 1e-4   # backward <= 1e-4 (atomic reordering)


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k].astype(np.float64) for k in z.files}


def case_by_name(name):
    for c in cases.SMALL_CASES + cases.MEDIUM_CASES + cases.DEPTHAWARE_CASES:
        if c["name"] == name:
            return c
    raise KeyError(name)
