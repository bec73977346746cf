"""CPU test: the two solvers that have no reference fixture at all -- the multi-keyframe local BA (north-star extension,
SURVEY a17 / D1) and the optimum of LocalBA::PoseOptimization (LocalBA.cpp:291-490, g2o absent from the image) -- against an
INDEPENDENT library: scipy.optimize.least_squares on the same reprojection residuals must reach the same optimum as
oracle.local_ba / oracle.pose_opt run to convergence (cost within 1e-8 relative, poses within 1e-6).

This does not pin parity with g2o (still no reference vector: "parity unpinned"); it shows that the oracle the GPU kernels are
compared with minimises the function it claims to minimise. The problems are built so that every edge is an inlier at the
optimum (chi2 far below the Huber bound 5.991): g2o's Huber kernel acts on the 2-vector error of an edge, scipy's `loss`
on scalar residuals, so only the plain least-squares optimum is comparable; the last case has gross outliers and checks the
optimum over the edges the oracle kept, which is what its last two non-robust rounds minimise (LocalBA.cpp:426-443)."""
import numpy as np
import pytest
from scipy.optimize import least_squares

import oracle
from trackingbench_slam_amd import synth

K = (718.856, 718.856, 607.1928, 185.2157)


def _rodrigues(w):
    th = np.linalg.norm(w)
    Wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + Wx
    return np.eye(3) + np.sin(th) / th * Wx + (1 - np.cos(th)) / th ** 2 * (Wx @ Wx)


def _as_solver_pose(T):
    """The solvers take float32 4x4 poses, whose rotation blocks are orthonormal only to ~6e-8, and work on the unit
    quaternion extracted from them (SE3Quat: oracle_pose.cpp:28-58,92-98). The same extraction here -- with the raw matrix the
    cost function differs by ~3e-7 relative, more than the agreement asked for."""
    R = T[:3, :3].astype(np.float64)
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        t = np.sqrt(t + 1.0)
        qw = 0.5 * t
        t = 0.5 / t
        q = np.array([(R[2, 1] - R[1, 2]) * t, (R[0, 2] - R[2, 0]) * t, (R[1, 0] - R[0, 1]) * t, qw])
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        c = np.zeros(3)
        c[i] = 0.5 * t
        t = 0.5 / t
        qw = (R[k, j] - R[j, k]) * t
        c[j] = (R[j, i] + R[i, j]) * t
        c[k] = (R[k, i] + R[i, k]) * t
        q = np.array([c[0], c[1], c[2], qw])
    x, y, z, w = q / np.linalg.norm(q)
    out = np.eye(4)
    out[:3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                   [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                   [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
    out[:3, 3] = T[:3, 3].astype(np.float64)
    return out


def _compose(xi, T0):
    """Tcw = [R(w) | t] * T0 -- any smooth chart works, the optimum is a point on SE(3)"""
    T = np.eye(4)
    T[:3, :3] = _rodrigues(xi[:3])
    T[:3, 3] = xi[3:6]
    return T @ T0


def _reproj(T, X, uv, w):
    pc = X @ T[:3, :3].T + T[:3, 3]
    proj = np.stack([K[0] * pc[:, 0] / pc[:, 2] + K[2], K[1] * pc[:, 1] / pc[:, 2] + K[3]], 1)
    return ((uv - proj) * np.sqrt(w)[:, None]).ravel()


def _solve(fun, x0):
    r = least_squares(fun, x0, jac="3-point", method="trf", x_scale="jac", ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=400)
    return r.x, 2.0 * r.cost


@pytest.mark.parametrize("seed,nkf,npt,nfixed", [(101, 4, 60, 2), (102, 6, 120, 2), (103, 5, 90, 1)])
def test_local_ba_optimum_matches_scipy(seed, nkf, npt, nfixed):
    Pt, Pi, Xt, Xi, obs = synth.ba_problem(seed, nkf, npt, K, obs_per_pt=4, noise_px=0.3, pose_noise=0.01, pt_noise=0.02)
    # a point seen by one keyframe has a free direction (its depth) and a residual that only vanishes in the limit: LM stops
    # a few 1e-7 of the cost short of it. Keep the points that two or more keyframes see, renumbered.
    cnt = np.bincount(obs["pt"], minlength=npt)
    keep = np.flatnonzero(cnt >= 2)
    remap = -np.ones(npt, np.int64); remap[keep] = np.arange(len(keep))
    obs = obs[cnt[obs["pt"]] >= 2].copy()
    obs["pt"] = remap[obs["pt"]]
    Xi, npt = Xi[keep], len(keep)
    if nfixed == 1:  # one fixed keyframe leaves the scale free: hold it through a second, exactly known pose instead
        Pi = Pi.copy(); Pi[1] = Pt[1]; nfixed_o = 2
    else:
        nfixed_o = nfixed
    it, Po, Xo, st = oracle.local_ba(K, Pi, nfixed_o, Xi, obs, 60)
    P0 = np.stack([_as_solver_pose(T) for T in Pi])
    X0 = Xi.astype(np.float64)
    kf, pt = obs["kf"], obs["pt"]
    uv = np.stack([obs["u"], obs["v"]], 1).astype(np.float64)
    w = obs["inv_sigma2"].astype(np.float64)
    nfree = nkf - nfixed_o

    def fun(x):
        T = [P0[k] if k < nfixed_o else _compose(x[6 * (k - nfixed_o):6 * (k - nfixed_o) + 6], P0[k]) for k in range(nkf)]
        X = X0 + x[6 * nfree:].reshape(-1, 3)
        r = np.empty(2 * len(obs))
        for k in range(nkf):
            m = kf == k
            r[np.repeat(m, 2)] = _reproj(T[k], X[pt[m]], uv[m], w[m])
        return r

    x, cost = _solve(fun, np.zeros(6 * nfree + 3 * npt))
    # every edge an inlier at the optimum, so the Huber sum the oracle reports IS the plain sum of squares
    r = fun(x).reshape(-1, 2)
    assert (r ** 2).sum(1).max() < 5.991
    assert abs(st[2] - cost) <= 1e-8 * cost, (st[2], cost)
    for k in range(nfixed_o, nkf):
        T = _compose(x[6 * (k - nfixed_o):6 * (k - nfixed_o) + 6], P0[k])
        assert np.abs(T - Po[k]).max() < 2e-6  # the oracle returns float32 poses
    seen = np.zeros(npt, bool); seen[pt] = True
    assert np.abs((X0 + x[6 * nfree:].reshape(-1, 3))[seen] - Xo[seen]).max() < 2e-5


@pytest.mark.parametrize("seed,n,outlier_frac", [(201, 120, 0.0), (202, 400, 0.0), (203, 300, 0.12)])
def test_pose_opt_optimum_matches_scipy(seed, n, outlier_frac):
    Tt, Ti, obs = synth.pose_problem(seed, n, K, noise_px=0.3, outlier_frac=outlier_frac)
    ninl, To, outl, st = oracle.pose_opt(K, Ti, obs)
    keep = outl == 0
    assert ninl == keep.sum() and (outlier_frac > 0) == (not keep.all())
    X = np.stack([obs["X"], obs["Y"], obs["Z"]], 1).astype(np.float64)[keep]
    uv = np.stack([obs["u"], obs["v"]], 1).astype(np.float64)[keep]
    w = obs["inv_sigma2"].astype(np.float64)[keep]
    T0 = _as_solver_pose(To)  # start at the oracle's answer: scipy must not move away from it ...
    x, cost = _solve(lambda x: _reproj(_compose(x, T0), X, uv, w), np.zeros(6))
    assert np.abs(_compose(x, T0) - To).max() < 2e-6
    # ... and from the initial pose it must arrive there as well (the optimum over the kept edges, not a local artefact)
    T1 = _as_solver_pose(Ti)
    x2, cost2 = _solve(lambda x: _reproj(_compose(x, T1), X, uv, w), np.zeros(6))
    assert np.abs(_compose(x2, T1) - To).max() < 2e-6
    assert abs(cost2 - cost) <= 1e-8 * cost
    c_or = (_reproj(T0, X, uv, w) ** 2).sum()
    assert abs(c_or - cost) <= 1e-6 * cost  # the float32 pose the oracle returns costs the same to first order
    if outlier_frac == 0:  # stats[1]: the chi2 the solver ended its last round with, all edges active
        assert abs(st[1] - cost) <= 1e-8 * cost, (st[1], cost)
