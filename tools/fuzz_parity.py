#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box: every operator of the C ABI against the oracle on seeded random
geometries, time bounded.  Not part of the test suite (minutes, not seconds); prints the first mismatch and exits 1.

    python tools/fuzz_parity.py [--seconds 240] [--seed 1]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import oracle  # noqa: E402
from trackingbench_slam_amd import capi, synth  # noqa: E402

K = (718.856, 718.856, 607.1928, 185.2157)


def same_rec(a, b):
    return len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in a.dtype.names)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    ctx = capi.Context(0)
    t0, it, counts = time.time(), 0, {}

    def fail(what, **info):
        print("MISMATCH in %s: %s" % (what, info))
        sys.exit(1)

    while time.time() - t0 < args.seconds:
        it += 1
        w, h = int(rng.integers(70, 900)), int(rng.integers(70, 700))
        nl, sc = int(rng.integers(2, 9)), float(rng.choice([0.5, 0.6, 0.7, 0.75, 0.8, 0.85, 0.9]))
        while nl > 2 and min(w, h) * sc ** (nl - 1) < 40:  # keep every level large enough for one 30-px cell + border
            nl -= 1
        img = synth.frame(int(rng.integers(0, 10 ** 6)), w, h)
        lv, sf = ctx.pyramid(img, nl, sc)
        olv, osf = oracle.pyramid(img, nl, sc)
        if not all(np.array_equal(a, b) for a, b in zip(lv, olv)):
            fail("pyramid", w=w, h=h, nl=nl, sc=sc)
        tgt, ith, mth = int(rng.integers(20, 2500)), int(rng.integers(15, 90)), int(rng.integers(5, 30))
        mth = min(mth, ith)
        k1, d1, _ = ctx.orb_extract(lv, sf, tgt, ith, mth)
        ko, do, _ = oracle.orb_extract(olv, sf, tgt, ith, mth)
        if not (np.array_equal(k1, ko) and np.array_equal(d1, do)):
            fail("orb_extract", w=w, h=h, nl=nl, sc=sc, tgt=tgt, ith=ith, mth=mth, n=(len(k1), len(ko)))
        th = int(rng.integers(5, 60))
        c1 = ctx.fast_detect(img, th)
        c2 = oracle.fast9(img, th)
        if not same_rec(c1, c2):
            fail("fast_detect", w=w, h=h, th=th)
        img2 = synth.frame(int(rng.integers(0, 10 ** 6)), w, h)
        lv2, _ = ctx.pyramid(img2, nl, sc)
        k2, d2, _ = ctx.orb_extract(lv2, sf, tgt, ith, mth)
        if len(k1) and len(k2):
            ratio, mt = float(rng.uniform(1.5, 12)), float(rng.uniform(20, 90))
            if not same_rec(ctx.search_by_bf(d1, d2, ratio, mt), oracle.search_by_bf(d1, d2, ratio, mt)):
                fail("search_by_bf", n1=len(k1), n2=len(k2), ratio=ratio, mt=mt)
            r, tl, nr = float(rng.uniform(3, 60)), int(rng.integers(30, 120)), float(rng.uniform(0.5, 1.0))
            a = (k1, d1, k2, d2, w, h, 0, nl, r, tl, nr, 30, bool(rng.integers(0, 2)))
            if not same_rec(ctx.search_by_violence(*a), oracle.search_by_violence(*a)):
                fail("search_by_violence", w=w, h=h, r=r, tl=tl, nr=nr)
        # optical flow (img -> img2 are unrelated frames: most tracks fail or wander, which is the interesting part) and CLAHE
        npt, ml = int(rng.integers(1, 400)), int(rng.integers(0, 6))
        pts = np.stack([rng.uniform(-30, w + 30, npt), rng.uniform(-30, h + 30, npt)], 1).astype(np.float32)
        if len(k1):
            pts[:min(npt, len(k1))] = np.stack([k1["x"], k1["y"]], 1)[:npt]
        shifted = np.roll(img, (int(rng.integers(-6, 7)), int(rng.integers(-9, 10))), (0, 1)) if it % 2 else img2
        fg, fo = ctx.optical_flow_pyr_lk(img, shifted, pts, max_level=ml), oracle.optical_flow_pyr_lk(img, shifted, pts, max_level=ml)
        if not (fg[3] == fo[3] and np.array_equal(fg[1], fo[1]) and np.array_equal(fg[0].view(np.uint32), fo[0].view(np.uint32)) and
                np.array_equal(fg[2].view(np.uint32), fo[2].view(np.uint32))):
            fail("optical_flow_pyr_lk", w=w, h=h, npt=npt, ml=ml, status=(int(fg[1].sum()), int(fo[1].sum())))
        # RANSAC fundamental matrix (rejectWithF), stereo-like point sets with a random share of gross outliers
        nr_ = int(rng.integers(7, 22)) if rng.integers(0, 5) == 0 else int(rng.integers(15, 2500))   # 8..14: the LMedS branch
        px = rng.uniform(10, w + 300, nr_); py = rng.uniform(10, h + 100, nr_)
        p1 = np.stack([px, py], 1).astype(np.float32)
        p2 = np.stack([px - 386.0 / rng.uniform(4, 60, nr_), py], 1) + rng.normal(0, float(rng.uniform(0, 0.6)), (nr_, 2))
        nbad = int(rng.integers(0, nr_))
        bad = rng.choice(nr_, nbad, replace=False)
        p2[bad] += rng.uniform(3, 90, (nbad, 2)) * rng.choice([-1, 1], (nbad, 2))
        p2 = p2.astype(np.float32)
        ro, rg = oracle.find_fundamental_ransac(p1, p2), ctx.find_fundamental_ransac(p1, p2)
        if not (ro[0] == rg[0] and ro[3] == rg[3] and np.array_equal(ro[1], rg[1]) and (not ro[0] or np.array_equal(ro[2].view(np.uint64), rg[2].view(np.uint64)))):
            fail("find_fundamental_ransac", n=nr_, nbad=nbad, iters=(rg[3], ro[3]), inliers=(int(rg[1].sum()), int(ro[1].sum())))
        st = (rng.uniform(size=nr_) < 0.8).astype(np.uint8)
        if True:
            if not np.array_equal(ctx.reject_with_f(p2, p1, st), oracle.reject_with_f(p2, p1, st)):
                fail("reject_with_f", n=nr_, live=int(st.sum()))
        if len(k1) >= 40 and it % 2 == 0:   # the whole optical-flow matcher with the RANSAC stage, and the stereo depths
            cam = oracle.camera(500, 500, w / 2, h / 2, w, h)
            keys = np.stack([k1["x"], k1["y"]], 1)[k1["octave"] == 0].astype(np.float32)
            try:
                eo = oracle.add_map_points_by_stereo(shifted, img, cam, keys, 386.1448)
            except oracle.OracleError:
                eo = None                    # 8..14 tracked points: OpenCV's LMedS branch, unsupported on both sides
            if eo is not None:
                eg, _ = ctx.add_map_points_by_stereo(shifted, img, cam, keys, 386.1448)
                if not np.array_equal(eg.view(np.uint32), eo.view(np.uint32)):
                    fail("add_map_points_by_stereo", w=w, h=h, n=len(keys), set=(int((eg > 0).sum()), int((eo > 0).sum())))
        # searchByBow on feature vectors made by a descriptor hash
        if len(k1) and len(k2):
            mod = int(rng.integers(1, 200))
            fv1, fv2 = {}, {}
            for i in range(len(d1)):
                fv1.setdefault(int(d1[i, 3] ^ d1[i, 17]) % mod, []).append(i)
            for i in rng.permutation(len(d2)):
                fv2.setdefault(int(d2[i, 3] ^ d2[i, 17]) % mod + (7 if i % 13 == 0 else 0), []).append(int(i))
            bw = dict(th_low=int(rng.integers(30, 260)), nratio=float(rng.uniform(0.5, 1.5)), histo_len=int(rng.choice([30, 45])),
                      check_orientation=bool(rng.integers(0, 2)), map_point_only=bool(rng.integers(0, 2)),
                      has_mp2=(rng.uniform(size=len(k2)) < 0.7).astype(np.uint8))
            if not same_rec(ctx.search_by_bow(k1, d1, fv1, k2, d2, fv2, **bw), oracle.search_by_bow(k1, d1, fv1, k2, d2, fv2, **bw)):
                fail("search_by_bow", n1=len(k1), n2=len(k2), mod=mod)
        # Frame::SetBow's transform on seeded vocabularies (regular and ragged trees), the extracted descriptors + near-word ones
        if it % 3 == 1 and len(d1):
            kv, Lv = int(rng.integers(2, 11)), int(rng.integers(1, 6))
            voc = synth.vocabulary(int(rng.integers(0, 10 ** 6)), kv, Lv, ragged=float(rng.choice([0.0, 0.0, 0.2, 0.4])))
            dd = np.concatenate([d1[: int(rng.integers(1, len(d1) + 1))], synth.descriptors_near_words(int(rng.integers(0, 10 ** 6)), voc, int(rng.integers(1, 300)))])
            lu = int(rng.integers(0, Lv + 2))
            hv = ctx.vocab_create(voc)
            try:
                gw = ctx.bow_transform(hv, dd, lu)
            finally:
                ctx.vocab_destroy(hv)
            ow = oracle.bow_transform(voc, dd, lu)
            if not (np.array_equal(gw[0], ow[0]) and np.array_equal(gw[1].view(np.uint64), np.asarray(ow[1], np.float64).view(np.uint64)) and np.array_equal(gw[2], ow[2])):
                fail("bow_transform", k=kv, L=Lv, levelsup=lu, n=len(dd))
        # FAST cell loop on dense images (the block kernel's list-free path): noise, low thresholds
        if it % 4 == 0:
            nw, nh = int(rng.integers(70, 400)), int(rng.integers(70, 300))
            noise = rng.integers(0, 256, (nh, nw), dtype=np.uint8)
            if rng.integers(0, 2):
                noise[:, : nw // 2] = synth.frame(int(rng.integers(0, 10 ** 6)), nw, nh)[:, : nw // 2]
            nlv, nsf = oracle.pyramid(noise, 2, 0.8)
            t1, t2 = int(rng.integers(0, 30)), int(rng.integers(0, 12))
            kg, dg, _ = ctx.orb_extract(nlv, nsf, 500, t1, t2)
            kq, dq, _ = oracle.orb_extract(nlv, nsf, 500, t1, t2)
            if not (np.array_equal(kg, kq) and np.array_equal(dg, dq)):
                fail("orb_extract on noise", w=nw, h=nh, t1=t1, t2=t2)
        clip, tiles = float(rng.choice([0.0, 0.5, 2.0, 3.0, 40.0])), (int(rng.integers(1, 10)), int(rng.integers(1, 10)))
        low = (img // int(rng.integers(1, 6)) + int(rng.integers(0, 60))).astype(np.uint8)
        if not np.array_equal(ctx.clahe(low, clip, tiles), oracle.clahe(low, clip, tiles)):
            fail("clahe", w=w, h=h, clip=clip, tiles=tiles)
        c = synth.projection_case(int(rng.integers(0, 10 ** 6)), n1=int(rng.integers(50, 2500)), nmp=int(rng.integers(50, 2500)))
        nrat = float(rng.uniform(1, 15))
        pa = (c["Tcw"], c["cam"], c["width"], c["height"], c["k1"], c["d1"], c["taken1"], c["k2"], c["mp"], c["mp_desc"], c["sf"], nrat)
        if not same_rec(ctx.search_by_projection(*pa), oracle.search_by_projection(*pa)):
            fail("search_by_projection", nrat=nrat)
        pm = (c["Tcw"], c["cam"], c["width"], c["height"], c["k1"], c["d1"], c["taken1"], c["mp"], c["mp_desc"], c["sf"], nrat, 0.8)
        if not same_rec(ctx.search_by_projection_map(*pm), oracle.search_by_projection_map(*pm)):
            fail("search_by_projection_map", nrat=nrat)
        n = int(rng.integers(10, 1500))
        _, Ti, obs = synth.pose_problem(int(rng.integers(0, 10 ** 6)), n, K)
        ng, Tg, og, _ = ctx.pose_opt(K, Ti, obs)
        no, To, oo, _ = oracle.pose_opt(K, Ti, obs)
        if not (ng == no and np.array_equal(og, oo) and np.allclose(Tg, To, rtol=1e-6, atol=1e-6)):
            fail("pose_opt", n=n, inl=(ng, no), dT=float(np.abs(Tg - To).max()))
        nkf, nfx = int(rng.integers(3, 13)), int(rng.integers(0, 3))
        if it % 3 == 0:  # a large window (11..64 free keyframes: the block-pair Schur kernel and the panel solve)
            nkf = int(rng.integers(13, 67))
        nfx = min(nfx, nkf - 1)
        if nkf - nfx <= 64:
            npt, per = int(rng.integers(20, 1500)), int(rng.integers(2, min(nkf, 14) + 1))
            far = rng.integers(0, 4) == 0   # a start far from the optimum: the LM loop rejects steps on the way
            bseed = int(rng.integers(0, 10 ** 6))
            bnoise = dict(pose_noise=float(rng.uniform(0.5, 3.0)), pt_noise=float(rng.uniform(2.0, 12.0))) if far else {}
            Pt, Pi, Xt, Xi, bo = synth.ba_problem(bseed, nkf, npt, K, obs_per_pt=per, **bnoise)
            itn = int(rng.integers(1, 8))
            ig, Pg, Xg, sg = ctx.local_ba(K, Pi, nfx, Xi, bo, itn)
            io, Po, Xo, so = oracle.local_ba(K, Pi, nfx, Xi, bo, itn)
            tol = 1e-6
            if not (np.allclose(Pg, Po, rtol=tol, atol=tol * max(1.0, float(np.abs(Po).max()))) and
                    np.allclose(Xg, Xo, rtol=tol, atol=tol * max(1.0, float(np.abs(Xo).max()))) and np.isclose(sg[2], so[2], rtol=1e-6, atol=1e-9)):
                # Far-from-optimum windows can hit an ill-conditioned LM step (a near-singular point block, a residual on the
                # Huber threshold): the CPU solver then moves by MORE than the tolerance when its float32 input points change by
                # one ulp, and no implementation can be expected to follow it closer than that. Such a case is counted, not
                # failed -- as long as the GPU is no further from the CPU solver than the CPU solver is from itself.
                dg = max(float(np.abs(Pg - Po).max()), float(np.abs(Xg - Xo).max()))
                ds = 0.0
                for toward in (1e9, -1e9):   # one ulp up, one ulp down: the movement depends on the direction
                    i2, P2, X2, s2 = oracle.local_ba(K, Pi, nfx, np.nextafter(Xi, np.float32(toward)), bo, itn)
                    ds = max(ds, float(np.abs(P2 - Po).max()), float(np.abs(X2 - Xo).max()))
                if far and dg <= 2.0 * ds:
                    counts["local_ba ill-conditioned (CPU solver moves as much at +-1 ulp of the input)"] = counts.get("local_ba ill-conditioned (CPU solver moves as much at +-1 ulp of the input)", 0) + 1
                    print("ill-conditioned local_ba case: seed %d nkf %d nfx %d npt %d itn %d: GPU vs CPU %.1e, CPU vs CPU at +-1 ulp %.1e" % (bseed, nkf, nfx, npt, itn, dg, ds), flush=True)
                else:
                    fail("local_ba", seed=bseed, noise=bnoise, nkf=nkf, nfx=nfx, npt=npt, per=per, itn=itn, dP=float(np.abs(Pg - Po).max()), dX=float(np.abs(Xg - Xo).max()),
                         chi=(float(sg[2]), float(so[2])), cpu_plus_1ulp=ds)
        if it % 10 == 0:
            print("iteration %d, %.0f s" % (it, time.time() - t0), flush=True)
    print("fuzz ok: %d iterations in %.0f s%s" % (it, time.time() - t0, "".join("; %s: %d" % kv for kv in counts.items())))
    ctx.close()


if __name__ == "__main__":
    main()
