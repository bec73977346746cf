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
    # ba_dims() in k_ba.hip: 2 x 256 workgroup slots = 2048 wavefronts over 171 windows = 11 each and 167 of them one more; one
    # partial system (lower triangle + rhs) per Schur wavefront
    nparts = 11 * 171 + 167
    assert nb == 171 * 20000 * 16 + 171 * 5000 * 96 + nparts * (np_ * (np_ + 1) // 2 + np_) * 8
    # a small batch: 20 workgroups per window, added through LDS to one partial system each
    _, nb8 = bench.schur_roofs(5000, 8, 10, free_edges=8 * 20000)
    assert nb8 == 8 * 20000 * 16 + 8 * 5000 * 96 + 8 * 20 * (np_ * (np_ + 1) // 2 + np_) * 8


def test_schur_sparse_flops_follow_survey_8d():
    # sum over points of k^2 * 216 per window and trial; 5000 points with 5 free observations each = 27 MFLOP
    assert bench.schur_flops_sparse(5000 * 25) == 27.0e6
    assert bench.extractor_bytes_per_image(1280, 720, 8, 0.8, 2000) == 15426725   # the 15.43 MB 'extractor total' row


def test_pmc_traffic_lookup(tmp_path, monkeypatch):
    # a file without units is a round-1 pass: 64 stereo frames (128 images) / 64 BA windows per launch
    doc = {"note": "x", "kernels": {"k_ba_schur": {"hbm_bytes_per_launch": 1000}, "k_resize_lds": {"hbm_bytes_per_launch": 640},
                                    "k_bf_nn": {"hbm_bytes_per_launch": 64}}}
    (tmp_path / "profiles").mkdir()
    (tmp_path / "profiles" / "r01_hbm_traffic_pmc.json").write_text(json.dumps(doc))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.pmc_traffic("k_ba_schur", 64) == 1000
    assert bench.pmc_traffic("k_ba_schur", 128) == 2000
    assert bench.pmc_traffic("k_resize", 128) == 640              # profile name of the kernel; units = images
    assert bench.pmc_traffic("k_bf_nn", 32) == 32                 # units = pairs
    assert bench.pmc_traffic("k_no_such_kernel", 64) is None
    assert bench.pmc_traffic("k_fast_cells", 64, (640, 480)) is None      # other geometries have their own passes or none
    d = bench.pmc_traffic("k_ba_schur", 170, detail=True)
    assert d["scaled"] and d["measured_at_units"] == 64 and d["file"].endswith("r01_hbm_traffic_pmc.json")
    # a newer pass that states its units wins and is used unscaled at those units
    doc2 = {"note": "y", "units_per_launch": {"images": 1024, "windows": 170, "pairs": 512},
            "kernels": {"k_ba_schur": {"hbm_bytes_per_launch": 5000}, "k_fast_cells": {"hbm_bytes_per_launch": 7}}}
    (tmp_path / "profiles" / "r02a_hbm_traffic_pmc.json").write_text(json.dumps(doc2))
    d = bench.pmc_traffic("k_ba_schur", 170, detail=True)
    assert d["bytes"] == 5000 and not d["scaled"]
    assert bench.pmc_traffic("k_fast_cells", 2048) == 14
    assert bench.pmc_traffic("k_resize", 128) == 640              # falls back to the older file for kernels the new one lacks


def test_committed_pmc_files_parse():
    assert bench.pmc_traffic("k_ba_schur", 64) > 0
    assert bench.pmc_traffic("k_ba_schur_pairs", 64, (3840, 2160)) > 0
