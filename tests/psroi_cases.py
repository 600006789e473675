"""Geometries and seeded inputs of the deformable PS-RoI pooling tests (CPU and GPU)."""
import numpy as np

# classes = offset.shape[1] // 2 (deform_pool_cuda_kernel.cu:289); C = out_channels * group_size^2
PSROI_CASES = [
    dict(name="rfcn_g3", B=2, out_channels=4, group_size=3, H=17, W=21, R=5, out_size=3, part_size=3,
         sample_per_part=4, trans_std=0.1, spatial_scale=0.31, classes=1, no_trans=False, seed=1),
    dict(name="pack_g1", B=2, out_channels=6, group_size=1, H=14, W=19, R=6, out_size=7, part_size=7,
         sample_per_part=2, trans_std=0.1, spatial_scale=0.27, classes=1, no_trans=False, seed=2),
    dict(name="two_classes_part4", B=1, out_channels=6, group_size=2, H=16, W=16, R=4, out_size=4, part_size=2,
         sample_per_part=3, trans_std=0.3, spatial_scale=0.23, classes=2, no_trans=False, seed=3),
    dict(name="no_trans", B=3, out_channels=5, group_size=2, H=12, W=15, R=7, out_size=5, part_size=5,
         sample_per_part=4, trans_std=0.0, spatial_scale=0.19, classes=1, no_trans=True, seed=4),
    dict(name="group7_out7", B=1, out_channels=2, group_size=7, H=20, W=24, R=3, out_size=7, part_size=7,
         sample_per_part=4, trans_std=0.1, spatial_scale=0.37, classes=1, no_trans=False, seed=6),
    dict(name="no_rois", B=1, out_channels=3, group_size=1, H=8, W=8, R=0, out_size=3, part_size=3,
         sample_per_part=2, trans_std=0.1, spatial_scale=0.25, classes=1, no_trans=False, seed=7),
]


def make_psroi_inputs(case, dtype, image_size=None):
    """-> data, rois, offset, grad_out, kwargs of psroi_forward/backward.

    RoIs are in image pixels with fractional corners away from .5 (so round-half rules cannot matter), some
    hang over the border (dropped / clamped samples) and one is degenerate (x2 < x1: the 0.1 minimum size)."""
    rng = np.random.default_rng(case["seed"])
    c = case
    C = c["out_channels"] * c["group_size"] ** 2
    data = rng.standard_normal((c["B"], C, c["H"], c["W"])).astype(dtype)
    iw, ih = c["W"] / c["spatial_scale"], c["H"] / c["spatial_scale"]
    rois = np.zeros((c["R"], 5), dtype=dtype)
    for r in range(c["R"]):
        x1, y1 = rng.uniform(0, 0.7 * iw), rng.uniform(0, 0.7 * ih)
        x2, y2 = x1 + rng.uniform(0.1 * iw, 0.6 * iw), y1 + rng.uniform(0.1 * ih, 0.6 * ih)
        if r == 2:
            x2 = x1 - 3.0  # degenerate
        box = np.floor(np.array([x1, y1, x2, y2])) + rng.uniform(0.05, 0.4, 4) * rng.choice([-1, 1], 4)
        rois[r] = [rng.integers(0, c["B"]), *np.maximum(box, 0.0)]
    tc = 2 * c["classes"]
    offset = (rng.standard_normal((c["R"], tc, c["part_size"], c["part_size"])) if not c["no_trans"]
              else np.zeros((0,))).astype(dtype)
    gout = rng.standard_normal((c["R"], c["out_channels"], c["out_size"], c["out_size"])).astype(dtype)
    # the reference's extension takes both scalars as C `float` (deform_pool_cuda.cpp:36-38): use values a float holds
    c = dict(c, spatial_scale=float(np.float32(c["spatial_scale"])), trans_std=float(np.float32(c["trans_std"])))
    kw = dict(spatial_scale=c["spatial_scale"], out_size=c["out_size"], out_channels=c["out_channels"],
              no_trans=c["no_trans"], group_size=c["group_size"], part_size=c["part_size"],
              sample_per_part=c["sample_per_part"], trans_std=c["trans_std"])
    return data, rois, offset, gout, kw
