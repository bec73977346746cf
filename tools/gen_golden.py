#!/usr/bin/env python3
"""Generate tests/golden/oracle_golden_v1.npz from the CPU oracle.

The reference's tests hold no golden vectors and the reference cannot be
built here (8c), so these vectors are produced by OUR oracle on the reference's own in-tree images
(data/left.png, data/right.png -> tests/golden/*.pgm) and on seeded synthetic inputs.  They pin
oracle<->HIP agreement and guard the oracle against regressions; they do NOT pin agreement with
genuine OpenCV / g2o ("parity unpinned").
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from trackingbench_slam_amd import synth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def read_pgm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P5"
        w, h = map(int, f.readline().split())
        f.readline()
        return np.frombuffer(f.read(), np.uint8).reshape(h, w).copy()


def sha(a):
    return np.frombuffer(hashlib.sha1(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def main():
    L = read_pgm(os.path.join(GOLDEN, "kitti00_left_1241x376.pgm"))
    R = read_pgm(os.path.join(GOLDEN, "kitti00_right_1241x376.pgm"))
    out = {}
    for tag, nl, N in (("c5", 5, 1000), ("c8", 8, 2000)):
        lvL, sf = oracle.pyramid(L, nl, 0.8)
        lvR, _ = oracle.pyramid(R, nl, 0.8)
        out[f"{tag}_sf"] = sf
        out[f"{tag}_pyr_sha_left"] = np.stack([sha(l) for l in lvL])
        out[f"{tag}_pyr_sha_right"] = np.stack([sha(l) for l in lvR])
        out[f"{tag}_sizes"] = np.array([l.shape[::-1] for l in lvL], np.int32)
        k1, d1, q = oracle.orb_extract(lvL, sf, N, 80, 30)
        k2, d2, _ = oracle.orb_extract(lvR, sf, N, 80, 30)
        out[f"{tag}_quotas"] = q
        out[f"{tag}_kps_left"], out[f"{tag}_desc_left"] = k1, d1
        out[f"{tag}_kps_right"], out[f"{tag}_desc_right"] = k2, d2
        out[f"{tag}_bf_all"] = oracle.bf_match(d1, d2, True)
        out[f"{tag}_bf_10_30"] = oracle.search_by_bf(d1, d2, 10, 30)
        out[f"{tag}_violence"] = oracle.search_by_violence(k1, d1, k2, d2, L.shape[1], L.shape[0], 0, nl, 50.0,
                                                          th_low=30, nratio=5.0, histo_len=30, check_orientation=True)
        if tag == "c5":
            for i, l in enumerate(lvL):
                out[f"c5_cand_left_L{i}"] = oracle.orb_candidates(l, 80, 30)
            out["c5_blur_sha_left"] = np.stack([sha(oracle.gaussian7(l)) for l in lvL])
            out["c5_fast9_th20_left_L0"] = oracle.fast9(lvL[0], 20, True)
            isf = oracle.scale_factors(nl, 0.8)[1]
            out["c5_fastgrid_left"] = oracle.fastgrid_extract(lvL, isf, 1000, 20.0)
            # AddPoints: second call with the first call's keypoints as exit keys
            ka, da, _ = oracle.orb_extract(lvL, sf, N, 80, 30, exit_keys=k1, quotas=q)
            out["c5_addpoints_kps_left"], out["c5_addpoints_desc_left"] = ka, da

    K = (718.856, 718.856, 607.1928, 185.2157)
    Tt, Ti, obs = synth.pose_problem(7, 200, K)
    n, T, outl, stats = oracle.pose_opt(K, Ti, obs)
    out["pose_obs"], out["pose_Tinit"], out["pose_Ttrue"] = obs, Ti, Tt
    out["pose_T"], out["pose_outlier"], out["pose_n"], out["pose_stats"] = T, outl, np.int32(n), stats

    Pt, Pi, Xt, Xi, bo = synth.ba_problem(3, 6, 300, K)
    it, P, X, st = oracle.local_ba(K, Pi, 2, Xi, bo, 10)
    out["ba_obs"], out["ba_poses_init"], out["ba_pts_init"] = bo, Pi, Xi
    out["ba_poses"], out["ba_pts"], out["ba_stats"] = P, X, st

    S = synth.frame(5, 640, 480)
    out["synth5_640x480_sha"] = sha(S)
    lv, sf = oracle.pyramid(S, 8, 0.8)
    ks, ds, _ = oracle.orb_extract(lv, sf, 1000, 80, 30)
    out["synth5_kps"], out["synth5_desc"] = ks, ds

    path = os.path.join(GOLDEN, "oracle_golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")
    large_ba()


def large_ba():
    """Known-answer vector of a large local-BA window (22 free keyframes: the block-pair Schur / panel-solve path of
    the HIP library), its own file so that oracle_golden_v1.npz stays as committed."""
    K = (718.856, 718.856, 607.1928, 185.2157)
    Pt, Pi, Xt, Xi, bo = synth.ba_problem(41, 24, 400, K, obs_per_pt=8)
    it, P, X, st = oracle.local_ba(K, Pi, 2, Xi, bo, 6)
    out = dict(ba_obs=bo, ba_poses_init=Pi, ba_pts_init=Xi, ba_poses=P, ba_pts=X, ba_stats=st)
    path = os.path.join(GOLDEN, "oracle_golden_ba_large_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
