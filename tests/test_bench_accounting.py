"""CPU tests of bench.py's roofline accounting (no GPU): the per-unit algorithmic bytes of SURVEY 8(d), the Schur
kernel's flop / byte model and the PMC-traffic lookup that fills `roofline.traffic`."""
import json
import os

import numpy as np

import bench


def test_level_pixels_match_survey_appendix_b():
    # 1280x720, 8 levels x0.8 -> sum of level pixels 2 484 905
    px = bench.level_pixels(1280, 720, 8, 0.8)
    assert px[0] == 1280 * 720 and sum(px) == 2484905
    assert all(a > b for a, b in zip(px, px[1:]))


def test_algorithmic_bytes_rows():
    w, h, nl, sc, tgt = 1280, 720, 8, 0.8, 2000
    px = bench.level_pixels(w, h, nl, sc)
    assert bench.algorithmic_bytes("k_fast_cells", w, h, nl, sc, tgt, 4, 2) == 4 * sum(px)
    assert bench.algorithmic_bytes("k_resize", w, h, nl, sc, tgt, 1, 1) == sum(px[i - 1] + px[i] for i in range(1, nl))
    assert bench.algorithmic_bytes("k_describe", w, h, nl, sc, tgt, 1, 1) == 2 * sum(px) + tgt * 2 * 961 + tgt * 60
    assert bench.algorithmic_bytes("k_bf_nn", w, h, nl, sc, tgt, 8, 4) == 2 * tgt * 32 * 4
    assert bench.algorithmic_bytes("k_octree", w, h, nl, sc, tgt, 8, 4) == 0       # no 8d row: latency bound


def test_schur_model():
    fl, nb = bench.schur_roofs(5000, 171, 10, free_edges=171 * 20000)
    np_ = 48
    assert fl == 2.0 * (np_ * (np_ + 1) / 2 + np_) * 3 * 5000 * 171
    G = min(max(1024 // 171, 1), max((1250 + 3) // 4, 1))
    assert nb == 171 * 20000 * 16 + 171 * (5000 * 96 + G * (6 * 256 + np_) * 8)


def test_pmc_traffic_lookup_scales_with_units():
    path = os.path.join(bench.ROOT, "profiles", "r01_hbm_traffic_pmc.json")
    rec = json.load(open(path))["kernels"]
    assert bench.pmc_traffic("k_ba_schur", 64) == rec["k_ba_schur"]["hbm_bytes_per_launch"]
    assert bench.pmc_traffic("k_ba_schur", 128) == 2 * rec["k_ba_schur"]["hbm_bytes_per_launch"]
    assert bench.pmc_traffic("k_resize", 64) == rec["k_resize_lds"]["hbm_bytes_per_launch"]      # profile name of the kernel
    assert bench.pmc_traffic("k_no_such_kernel", 64) is None
    # other geometries have their own passes or none
    assert bench.pmc_traffic("k_ba_schur_pairs", 64, (3840, 2160)) > 0
    assert bench.pmc_traffic("k_fast_cells", 64, (640, 480)) is None
