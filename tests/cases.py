"""Shared parity cases: geometry sweep + seeded input generation (numpy RNG, fp64 master copies)."""
import numpy as np


def out_hw(H, W, kh, kw, s, p, d):
    return (H + 2 * p - (d * (kh - 1) + 1)) // s + 1, (W + 2 * p - (d * (kw - 1) + 1)) // s + 1


def C(name, N=2, Ci=4, Co=4, H=4, W=4, kh=3, kw=3, stride=1, pad=1, dil=1, groups=1, dg=1, bias=False,
      modulated=True, off_scale=2.0, special=None, seed=0):
    return dict(name=name, N=N, Ci=Ci, Co=Co, H=H, W=W, kh=kh, kw=kw, stride=stride, pad=pad, dil=dil,
                groups=groups, dg=dg, bias=bias, modulated=modulated, off_scale=off_scale, special=special,
                seed=seed)


# N=2, C in {4,8}, HxW in {4x4,7x5,9x9}, k=3 (+1x1, 3x1), stride {1,2}, dilation {1,2,4}
# with pad=dilation, groups {1,2}, dg {1,2,4}, bias on/off, offsets randn*2 + hand-placed edge cases.
SMALL_CASES = [
    C("ref_selftest_shape", Ci=4, Co=4, H=4, W=4, groups=2, bias=True, modulated=False, seed=3),
    C("ref_selftest_shape_mod", Ci=4, Co=4, H=4, W=4, groups=2, bias=True, modulated=True, seed=4),
    C("v1_7x5_d2", Ci=8, Co=6, H=7, W=5, pad=2, dil=2, modulated=False, seed=5),
    C("mod_7x5_d2_dg2", Ci=8, Co=6, H=7, W=5, pad=2, dil=2, dg=2, bias=True, seed=6),
    C("mod_9x9_d4_dg4_g2", Ci=8, Co=8, H=9, W=9, pad=4, dil=4, dg=4, groups=2, seed=7),
    C("mod_9x9_s2", Ci=4, Co=8, H=9, W=9, stride=2, pad=1, dil=1, bias=True, seed=8),
    C("v1_9x9_s2_d2", Ci=4, Co=4, H=9, W=9, stride=2, pad=2, dil=2, dg=2, modulated=False, bias=True, seed=9),
    C("mod_1x1", Ci=8, Co=4, H=7, W=5, kh=1, kw=1, pad=0, dil=1, seed=10),
    C("mod_3x1", Ci=4, Co=4, H=7, W=5, kh=3, kw=1, pad=0, dil=1, dg=2, seed=11),
    C("mod_pad0", Ci=4, Co=4, H=9, W=9, pad=0, dil=1, seed=12),
    C("mod_scattered", Ci=4, Co=4, H=9, W=9, pad=2, dil=2, off_scale=8.0, seed=13),
    C("mod_zero_offset", Ci=4, Co=4, H=7, W=5, pad=1, dil=1, off_scale=0.0, special="integer", seed=14),
    C("mod_edge_coords", Ci=4, Co=4, H=7, W=5, pad=1, dil=1, special="edges", seed=15),
    C("v1_edge_coords", Ci=4, Co=4, H=7, W=5, pad=1, dil=1, modulated=False, special="edges", seed=16),
    C("mod_mask01", Ci=4, Co=4, H=7, W=5, pad=1, dil=1, special="mask01", seed=17),
    C("mod_n1_odd", N=1, Ci=6, Co=5, H=11, W=13, pad=2, dil=2, dg=3, bias=True, seed=18),
    C("mod_bigk", Ci=4, Co=4, H=9, W=9, kh=5, kw=5, pad=2, dil=1, seed=19),
]

# Larger cases (still seconds on the CPU oracle) that exercise full GEMM tiles, K/M/N edges, batching.
MEDIUM_CASES = [
    C("med_mod_c64_33x29_d2", N=2, Ci=64, Co=48, H=33, W=29, pad=2, dil=2, seed=30),
    C("med_v1_c32_40x40_g2", N=3, Ci=32, Co=64, H=40, W=40, pad=1, dil=1, groups=2, dg=2, bias=True,
      modulated=False, seed=31),
    C("med_mod_c128_25x25_d4_dg4", N=2, Ci=128, Co=136, H=25, W=25, pad=4, dil=4, dg=4, bias=True, seed=32),
    C("med_mod_s2_c32_41x37", N=2, Ci=32, Co=32, H=41, W=37, stride=2, pad=1, dil=1, seed=33),
]

DEPTHAWARE_CASES = [
    C("da_4x4_g2", Ci=4, Co=4, H=4, W=4, groups=2, bias=True, seed=40),
    C("da_7x5_d2", Ci=8, Co=6, H=7, W=5, pad=2, dil=2, seed=41),
    C("da_9x9_s2", Ci=4, Co=8, H=9, W=9, stride=2, pad=1, dil=1, bias=True, seed=42),
    C("da_pad3_d1", Ci=4, Co=4, H=7, W=7, pad=3, dil=1, seed=43),   # centre tap can fall out of bounds
    C("da_9x9_d4", Ci=8, Co=8, H=9, W=9, pad=4, dil=4, groups=2, seed=44),
]


def geometry(c):
    Ho, Wo = out_hw(c["H"], c["W"], c["kh"], c["kw"], c["stride"], c["pad"], c["dil"])
    return Ho, Wo, c["kh"] * c["kw"]


def make_inputs(c):
    """fp64 numpy inputs for a case (deterministic)."""
    rng = np.random.default_rng(c["seed"])
    N, Ci, Co, H, W = c["N"], c["Ci"], c["Co"], c["H"], c["W"]
    Ho, Wo, K2 = geometry(c)
    x = rng.standard_normal((N, Ci, H, W))
    weight = rng.uniform(-1, 1, (Co, Ci // c["groups"], c["kh"], c["kw"])) / np.sqrt(Ci * K2)
    bias = rng.standard_normal((Co,)) if c["bias"] else None
    # Offsets are quantised to odd multiples of 2^-11 (never an integer) so that (integer tap position + offset) is exact in fp32:
    # d/d(offset) of bilinear sampling is discontinuous at integer coordinates, and an fp32 rounding of the sum
    # across such a boundary would make an fp32 implementation and the fp64 oracle pick different cells.
    offset = (np.floor(rng.standard_normal((N, c["dg"] * 2 * K2, Ho, Wo)) * c["off_scale"] * 1024.0) + 0.5) / 1024.0
    mask = 1.0 / (1.0 + np.exp(-rng.standard_normal((N, c["dg"] * K2, Ho, Wo)))) if c["modulated"] else None
    gout = rng.standard_normal((N, Co, Ho, Wo))
    depth = rng.standard_normal((N, 1, H, W))
    sp = c["special"]
    if c["off_scale"] == 0.0:
        offset[:] = 0.0
    if sp == "edges":
        # hand-placed coordinates: the sample position is base + offset; drive it to special values
        off = offset.reshape(N, c["dg"], K2, 2, Ho, Wo)
        base_h = (np.arange(Ho) * c["stride"] - c["pad"]).reshape(1, 1, 1, Ho, 1) + \
                 (np.arange(K2) // c["kw"] * c["dil"]).reshape(1, 1, K2, 1, 1)
        base_w = (np.arange(Wo) * c["stride"] - c["pad"]).reshape(1, 1, 1, 1, Wo) + \
                 (np.arange(K2) % c["kw"] * c["dil"]).reshape(1, 1, K2, 1, 1)
        targets = np.array([-1.0, -0.5, -1023.0 / 1024.0, -1.5, -3.25, 0.0, 2.0, H - 1.0, H - 0.5, float(H), H + 2.5,
                            1.25, 0.75, -0.25])
        th = targets[rng.integers(0, len(targets), size=(N, c["dg"], K2, Ho, Wo))]
        tw = targets[rng.integers(0, len(targets), size=(N, c["dg"], K2, Ho, Wo))]
        tw = np.where(tw >= H - 1, tw - H + W, tw)   # re-target the upper edge to the width
        pick = rng.random((N, c["dg"], K2, Ho, Wo)) < 0.7
        off[:, :, :, 0] = np.where(pick, th - base_h, off[:, :, :, 0])
        off[:, :, :, 1] = np.where(pick, tw - base_w, off[:, :, :, 1])
        offset = off.reshape(N, c["dg"] * 2 * K2, Ho, Wo)
    if sp == "mask01" and mask is not None:
        mask = (rng.random(mask.shape) < 0.5).astype(np.float64)
    return dict(x=x, weight=weight, bias=bias, offset=offset, mask=mask, gout=gout, depth=depth)


def kwargs(c):
    return dict(stride=c["stride"], padding=c["pad"], dilation=c["dil"], groups=c["groups"],
                deformable_groups=c["dg"])


def has_integer_coords(c):
    """Cases where sample coordinates sit exactly on the integer grid: d/d(offset) is one-sided there,
    so an autograd formulation may legitimately pick the other side."""
    return c["special"] in ("integer", "edges")


def relerr(a, b):
    """rel-err := max|a-b| / max|b| per tensor."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = np.abs(b).max()
    if denom == 0:
        return float(np.abs(a - b).max())
    return float(np.abs(a - b).max() / denom)
