"""CPU checks of bench.py's host-side arithmetic and bookkeeping (no GPU, no kernels): the SURVEY.md 8(d) per-image work
figures the bench line is priced with, the provenance rule of the HBM traffic figures (a committed counter file is used only
while the kernel sources it was measured on are the ones in the tree), and the scatter's LDS-atomic floor."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

W1 = dict(N=2, C=512, Co=512, H=97, W=97, k=3, stride=1, pad=4, dil=4, groups=1, dg=1)     # BASELINE configs[0] layer


def test_algorithmic_work_of_config_1_matches_survey_8d():
    a = bench.alg_numbers(W1)
    P, C = 97 * 97, 512
    assert a["b_im2col"] == 4 * (C * P + 27 * P + C * 9 * P) == 193_712_492          # 193.7 MB per image
    assert a["b_fused"] == 4 * (C * 9 * P + 2 * C * P + 54 * P) == 213_998_296       # fused col2im + coord: 214.0 MB
    assert a["f_gemm"] == 2.0 * 512 * 512 * 9 * P                                    # 44.39 GFLOP per GEMM and image
    assert abs(a["f_gemm"] / 1e9 - 44.39) < 0.01
    assert a["b_total"] == 2 * a["b_im2col"] + a["b_col2im"] + a["b_coord"] and a["f_total"] == 3 * a["f_gemm"]


def test_lds_atomic_floor_of_the_scatter(monkeypatch):
    monkeypatch.delenv("DDCN_SCATTER_FX", raising=False)
    f = bench.lds_atomic_floor(W1, 0.31)
    # 4 atomics per (pixel, tap, channel), 64 lanes per wave instruction, two channels per ds_add_u64
    assert f["wave_instructions"] == 2 * 512 * 97 * 97 * 9 * 4 // 64 // 2 == 2_709_792
    assert abs(f["floor_ms"] - 2_709_792 * 19 / (256 * 2.4e9) * 1e3) < 1e-4 and 0.08 < f["floor_ms"] < 0.09
    assert abs(f["frac"] - f["floor_ms"] / 0.31) < 1e-3
    monkeypatch.setenv("DDCN_SCATTER_FX", "0")
    g = bench.lds_atomic_floor(W1, 0.35)
    assert g["wave_instructions"] == 2 * f["wave_instructions"] and "1 channel" in g["format"]


def test_committed_traffic_is_tied_to_the_kernel_sources(monkeypatch):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
    assert files, "no committed counter file"
    rec = json.load(open(files[-1]))
    traffic, src = bench.committed_traffic(W1)
    if rec["csrc_sha256"] == bench.csrc_digest():
        assert set(traffic) == {"im2col", "gemm_out", "gemm_gcol", "col2im_coord", "gemm_gweight"} and "why" not in src
        # counters are an upper bound of what the algorithm has to move, not below it by more than the cache can explain
        a = bench.alg_numbers(W1)
        assert traffic["im2col"] > 0.9 * a["b_im2col"] * W1["N"]
    else:
        assert traffic == {} and "stale" in src["why"]
    # sources changed since the counters were collected -> the figure is dropped, not reported
    monkeypatch.setattr(bench, "csrc_digest", lambda: "0" * 64)
    traffic, src = bench.committed_traffic(W1)
    assert traffic == {} and "stale" in src["why"]
    # another workload than the one the passes were recorded for -> no figure either
    traffic, src = bench.committed_traffic(dict(W1, N=8))
    assert traffic == {} and src["file"] is None
