"""GPU parity tests (pytest -m gpu): multi-keyframe local BA (north-star extension with NO reference
counterpart) against this repo's FP64 CPU solver; tolerance 1e-6 relative on poses / points / chi2
(BASELINE.json north_star: "BA pose and reprojection error within 1e-6 relative")."""
import numpy as np
import pytest

import oracle
from trackingbench_slam_amd import capi, synth

pytestmark = pytest.mark.gpu
K = (718.856, 718.856, 607.1928, 185.2157)


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def _close(a, b, tol=1e-6):
    assert np.allclose(a, b, rtol=tol, atol=tol * max(1.0, float(np.abs(b).max()))), float(np.abs(a - b).max())


def test_local_ba_kat(ctx, golden):
    it, P, X, st = ctx.local_ba(K, golden["ba_poses_init"], 2, golden["ba_pts_init"], golden["ba_obs"], 10)
    _close(P, golden["ba_poses"])
    _close(X, golden["ba_pts"])
    assert np.isclose(st[2], golden["ba_stats"][2], rtol=1e-6)
    assert np.isclose(st[1], golden["ba_stats"][1], rtol=1e-9)
    assert it == int(golden["ba_stats"][0])


def test_local_ba_large_window_kat(ctx):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_golden_ba_large_v1.npz"), allow_pickle=False)
    it, P, X, st = ctx.local_ba(K, g["ba_poses_init"], 2, g["ba_pts_init"], g["ba_obs"], 6)
    _close(P, g["ba_poses"])
    _close(X, g["ba_pts"])
    assert np.isclose(st[2], g["ba_stats"][2], rtol=1e-6) and np.isclose(st[1], g["ba_stats"][1], rtol=1e-9)
    assert it == int(g["ba_stats"][0])


@pytest.mark.parametrize("seed,nkf,npt,nfixed,iters", [(1, 5, 200, 2, 10), (2, 10, 5000, 2, 10), (3, 3, 50, 1, 5),
                                                        (4, 10, 1000, 0, 10), (5, 12, 700, 2, 3), (6, 4, 33, 2, 10),
                                                        (7, 5, 9000, 2, 4)])   # > 8192 points: no renumbering, pattern sort through global buffers
def test_local_ba_vs_cpu_solver(ctx, seed, nkf, npt, nfixed, iters):
    Pt, Pi, Xt, Xi, obs = synth.ba_problem(seed, nkf, npt, K)
    rng = np.random.default_rng(seed)
    obs = obs[rng.permutation(len(obs))]  # the host entry point must not depend on the caller's order
    io, Po, Xo, so = oracle.local_ba(K, Pi, nfixed, Xi, obs, iters)
    ig, Pg, Xg, sg = ctx.local_ba(K, Pi, nfixed, Xi, obs, iters)
    _close(Pg, Po)
    _close(Xg, Xo)
    assert np.isclose(sg[2], so[2], rtol=1e-6, atol=1e-9) and np.isclose(sg[1], so[1], rtol=1e-9)
    assert np.abs(Pg[:nfixed] - Pi[:nfixed]).max() < 1e-6 if nfixed else True
    assert sg[2] < sg[1]


@pytest.mark.parametrize("seed,nkf,npt,nfixed,per_pt", [(11, 10, 600, 2, 10),  # every keyframe sees every point
                                                         (12, 6, 300, 1, 6), (13, 4, 120, 1, 4), (14, 3, 90, 2, 3),
                                                         (15, 12, 150, 2, 12), (16, 10, 203, 0, 2),
                                                         (17, 5, 3000, 1, 2),     # few patterns, hundreds of points each: full groups
                                                         (18, 10, 2500, 2, 7),    # patterns of 1..7 keyframes: both product paths
                                                         (19, 9, 900, 1, 9), (20, 8, 64, 1, 6), (21, 2, 500, 1, 2)])
def test_local_ba_chunk_capacities(ctx, seed, nkf, npt, nfixed, per_pt):
    """The Schur kernel sorts a window's points by visibility pattern and works on groups of one pattern (one lane per
    free-keyframe edge, up to 64): patterns of up to five keyframes run the compact v_mfma_f64_4x4x4 product, larger ones the
    direct vector form, and the kernel has one instance per R = ceil(6 nfree / 16). Sparse to fully dense windows with
    1..10 free keyframes walk through every instance, every pattern size and both group limits (points, edges)."""
    Pt, Pi, Xt, Xi, obs = synth.ba_problem(seed, nkf, npt, K, obs_per_pt=per_pt)
    io, Po, Xo, so = oracle.local_ba(K, Pi, nfixed, Xi, obs, 6)
    ig, Pg, Xg, sg = ctx.local_ba(K, Pi, nfixed, Xi, obs, 6)
    _close(Pg, Po)
    _close(Xg, Xo)
    assert np.isclose(sg[2], so[2], rtol=1e-6, atol=1e-9) and np.isclose(sg[1], so[1], rtol=1e-9)


@pytest.mark.parametrize("seed,nkf,npt,nfixed,per_pt,iters", [(31, 13, 300, 2, 6, 6),    # 11 free: first large-window size
                                                               (32, 50, 2000, 2, 8, 5),   # BASELINE configs[4]: 50-KF window
                                                               (33, 66, 800, 2, 12, 4),   # 64 free keyframes: the maximum
                                                               (34, 30, 400, 0, 30, 4),   # dense: every keyframe sees every point
                                                               (35, 20, 37, 1, 3, 10)])
def test_local_ba_large_window(ctx, seed, nkf, npt, nfixed, per_pt, iters):
    """More than 10 free keyframes (reduced system up to 384 x 384) take the generic Schur / solve kernels."""
    Pt, Pi, Xt, Xi, obs = synth.ba_problem(seed, nkf, npt, K, obs_per_pt=per_pt)
    io, Po, Xo, so = oracle.local_ba(K, Pi, nfixed, Xi, obs, iters)
    ig, Pg, Xg, sg = ctx.local_ba(K, Pi, nfixed, Xi, obs, iters)
    _close(Pg, Po)
    _close(Xg, Xo)
    assert np.isclose(sg[2], so[2], rtol=1e-6, atol=1e-9) and np.isclose(sg[1], so[1], rtol=1e-9)
    assert ig == io and sg[2] < sg[1]


def test_local_ba_large_window_batch(ctx):
    import torch
    from trackingbench_slam_amd.ba import BatchedLocalBA
    ba = BatchedLocalBA(ctx, 3, nkf=24, npt=500, iters=5, seed=5, device=torch.device("cuda", 0), distinct=3)
    ba.run()
    torch.cuda.synchronize()
    P1 = ba.poses.cpu().numpy().copy()
    ba.run()
    torch.cuda.synchronize()
    assert np.array_equal(P1, ba.poses.cpu().numpy())  # deterministic
    for w in range(3):
        n = int(ba.host["counts"][w])
        io, Po, Xo, so = oracle.local_ba(K, ba.host["poses"][w], 2, ba.host["pts"][w], ba.host["obs"][w, :n], 5)
        _close(P1[w].reshape(-1, 4, 4), Po)
        _close(ba.pts[w].cpu().numpy(), Xo)


def test_local_ba_batched_windows(ctx):
    import torch
    from trackingbench_slam_amd.ba import BatchedLocalBA
    ba = BatchedLocalBA(ctx, 5, nkf=6, npt=400, iters=8, seed=3, device=torch.device("cuda", 0), distinct=3)
    ba.run()
    ba.run()  # re-running from the stored initial state must reproduce the result bit for bit
    torch.cuda.synchronize()
    P1 = ba.poses.cpu().numpy().copy()
    ba.run()
    torch.cuda.synchronize()
    assert np.array_equal(P1, ba.poses.cpu().numpy())
    for w in range(5):
        n = int(ba.host["counts"][w])
        io, Po, Xo, so = oracle.local_ba(K, ba.host["poses"][w], 2, ba.host["pts"][w], ba.host["obs"][w, :n], 8)
        _close(P1[w].reshape(-1, 4, 4), Po)
        _close(ba.pts[w].cpu().numpy(), Xo)
        assert np.isclose(float(ba.stats[w, 2]), so[2], rtol=1e-6)


def test_local_ba_bench_launch_shape(ctx):
    """One of bench.py's three BA partitions: 171 windows of 10 keyframes / 5000 points per call -- the number of Schur
    workgroups per window (1024 / W) and with it the number of partial systems the solve adds up follow from W, so the
    shape the bench times gets its own check. 3 distinct windows tiled over the 171, each against the oracle."""
    import torch
    from trackingbench_slam_amd.ba import BatchedLocalBA
    ba = BatchedLocalBA(ctx, 171, nkf=10, npt=5000, iters=10, seed=9, device=torch.device("cuda", 0), distinct=3)
    ba.run()
    torch.cuda.synchronize()
    P = ba.poses.cpu().numpy(); X = ba.pts.cpu().numpy(); st = ba.stats.cpu().numpy()
    ref = []
    for w in range(3):
        n = int(ba.host["counts"][w])
        ref.append(oracle.local_ba(K, ba.host["poses"][w], 2, ba.host["pts"][w], ba.host["obs"][w, :n], 10))
    for w in range(171):
        io, Po, Xo, so = ref[w % 3]
        _close(P[w].reshape(-1, 4, 4), Po)
        _close(X[w], Xo)
        assert np.isclose(st[w, 2], so[2], rtol=1e-6) and int(st[w, 0]) == io
    assert np.array_equal(P[0], P[3]) and np.array_equal(X[1], X[4])      # copies of one window agree bit for bit


@pytest.mark.parametrize("W,peers", [(33, 1), (100, 1), (128, 1), (130, 1), (515, 1), (171, 3), (64, 2)])
def test_local_ba_batch_shapes(W, peers):
    """How the Schur launch deals its wavefronts depends on the batch: whole workgroups per window summed through LDS (small
    batches: 33, 64 and 100 windows), a resident round of wavefronts dealt one by one (128: sixteen each; 130: fifteen or
    sixteen), four per window when there are more windows than workgroup slots (515, or a context that shares the GPU:
    tb_set_concurrency). Every window of every shape against the CPU solver; copies of one window agree bit for bit."""
    import torch
    from trackingbench_slam_amd.ba import BatchedLocalBA
    c = capi.Context(0)
    try:
        c.set_concurrency(peers)
        ba = BatchedLocalBA(c, W, nkf=7, npt=700, iters=4, seed=17, device=torch.device("cuda", 0), distinct=3)
        ba.run()
        torch.cuda.synchronize()
        P = ba.poses.cpu().numpy(); X = ba.pts.cpu().numpy(); st = ba.stats.cpu().numpy()
        ref = []
        for w in range(3):
            n = int(ba.host["counts"][w])
            ref.append(oracle.local_ba(K, ba.host["poses"][w], 2, ba.host["pts"][w], ba.host["obs"][w, :n], 4))
        for w in range(W):
            io, Po, Xo, so = ref[w % 3]
            _close(P[w].reshape(-1, 4, 4), Po)
            _close(X[w], Xo)
            assert np.isclose(st[w, 2], so[2], rtol=1e-6) and int(st[w, 0]) == io
        assert np.array_equal(P[0], P[3 * ((W - 1) // 3)]) and np.array_equal(X[1], X[1 + 3 * ((W - 2) // 3)])
    finally:
        c.close()


@pytest.mark.parametrize("seed,nkf,npt,per,pose_noise,pt_noise", [(50, 3, 40, 3, 3.0, 12.0), (53, 5, 100, 3, 2.0, 10.0), (58, 3, 40, 3, 3.0, 12.0),
                                                                    (53, 3, 40, 3, 3.0, 12.0)])
def test_local_ba_rejected_steps(ctx, seed, nkf, npt, per, pose_noise, pt_noise):
    """Starts far from the optimum: the CPU solver rejects one to seven LM steps on the way (stats[4] of the oracle counts the
    trials). A rejected step keeps the linearisation and changes lambda only: the point records are rebuilt by k_ba_points
    from the stored blocks -- same trajectory, same number of iterations, same result as the CPU solver."""
    Pt, Pi, Xt, Xi, obs = synth.ba_problem(seed, nkf, npt, K, obs_per_pt=per, pose_noise=pose_noise, pt_noise=pt_noise)
    io, Po, Xo, so = oracle.local_ba(K, Pi, 1, Xi, obs, 8)
    assert so[4] > io, "the case is meant to contain rejected steps"
    ig, Pg, Xg, sg = ctx.local_ba(K, Pi, 1, Xi, obs, 8)
    assert ig == io
    _close(Pg, Po)
    _close(Xg, Xo)
    assert np.isclose(sg[2], so[2], rtol=1e-6, atol=1e-9) and np.isclose(sg[1], so[1], rtol=1e-9)


def test_local_ba_rejects_bad_input(ctx):
    Pt, Pi, Xt, Xi, obs = synth.ba_problem(1, 4, 30, K)
    bad = obs.copy(); bad["kf"][0] = 99
    with pytest.raises(capi.TBError):
        ctx.local_ba(K, Pi, 2, Xi, bad, 5)
    with pytest.raises(capi.TBError):  # a point seen twice by one keyframe
        ctx.local_ba(K, Pi, 2, Xi, np.concatenate([obs, obs[5:6]]), 5)
    with pytest.raises(capi.TBError) as e:  # more free keyframes than the 6-bit free-edge key holds
        P2 = np.tile(np.eye(4, dtype=np.float32), (70, 1, 1))
        ctx.local_ba(K, P2, 2, Xi, obs, 5)
    assert e.value.code == capi.TB_EUNSUPPORTED


def test_local_ba_unobserved_points_and_empty_window(ctx):
    """Points without any observation keep their position (their block is lambda I, their gradient zero); a window without
    observations is left untouched. Both next to ordinary windows in one batch."""
    Pt, Pi, Xt, Xi, obs = synth.ba_problem(21, 5, 60, K)
    o2 = obs[(obs["pt"] != 3) & (obs["pt"] != 7) & (obs["pt"] != 59)]
    io, Po, Xo, so = oracle.local_ba(K, Pi, 2, Xi, o2, 5)
    ig, Pg, Xg, sg = ctx.local_ba(K, Pi, 2, Xi, o2, 5)
    _close(Pg, Po)
    _close(Xg, Xo)
    assert np.array_equal(Xg[[3, 7, 59]], Xi[[3, 7, 59]])
    # only fixed keyframes observe a point: no Schur contribution, but it still moves
    o3 = obs[~((obs["pt"] == 5) & (obs["kf"] >= 2))]
    io, Po, Xo, so = oracle.local_ba(K, Pi, 2, Xi, o3, 5)
    ig, Pg, Xg, sg = ctx.local_ba(K, Pi, 2, Xi, o3, 5)
    _close(Pg, Po)
    _close(Xg, Xo)
    import torch
    from trackingbench_slam_amd.ba import BatchedLocalBA
    ba = BatchedLocalBA(ctx, 3, nkf=6, npt=200, iters=6, seed=9, device=torch.device("cuda", 0), distinct=3)
    ba.counts[1] = 0                      # the middle window has no observations
    ba.run()
    torch.cuda.synchronize()
    # (poses pass through the solver's unit quaternion and back: equal to rounding, not bit for bit)
    assert torch.allclose(ba.poses[1], ba.poses0[1], rtol=0, atol=1e-6) and torch.equal(ba.pts[1], ba.pts0[1])
    for w in (0, 2):
        n = int(ba.host["counts"][w])
        io, Po, Xo, so = oracle.local_ba(K, ba.host["poses"][w], 2, ba.host["pts"][w], ba.host["obs"][w, :n], 6)
        _close(ba.poses[w].cpu().numpy().reshape(-1, 4, 4), Po)
        _close(ba.pts[w].cpu().numpy(), Xo)


def test_local_ba_large_window_mixed_batch(ctx):
    """Large windows (18 free keyframes) side by side in one batch: an ordinary one, one without observations (left as it
    is), one with a repeated (keyframe, point) pair (flagged in stats[7], left as it is), one with unobserved points."""
    import torch
    from trackingbench_slam_amd.ba import BatchedLocalBA
    ba = BatchedLocalBA(ctx, 4, nkf=20, npt=300, iters=5, seed=7, device=torch.device("cuda", 0), distinct=4)
    obs = ba.host["obs"].copy()
    cnt = ba.host["counts"].copy()
    cnt[1] = 0
    obs[2, 11] = obs[2, 10]                                     # same (kf, pt) twice
    keep = ~np.isin(obs[3, :cnt[3]]["pt"], (0, 17, 299))        # three points nobody sees
    o3 = obs[3, :cnt[3]][keep]
    obs[3, :len(o3)] = o3
    cnt[3] = len(o3)
    ba.obs.copy_(torch.from_numpy(obs.view(np.uint8).reshape(ba.obs.shape)))
    ba.counts.copy_(torch.from_numpy(cnt))
    ba.run()
    torch.cuda.synchronize()
    st = ba.stats.cpu().numpy()
    assert st[2, 7] == -1 and st[0, 7] != -1 and st[3, 7] != -1
    for w in (1, 2):
        assert torch.allclose(ba.poses[w], ba.poses0[w], rtol=0, atol=1e-6) and torch.equal(ba.pts[w], ba.pts0[w])
    for w, o in ((0, obs[0, :cnt[0]]), (3, o3)):
        io, Po, Xo, so = oracle.local_ba(K, ba.host["poses"][w], 2, ba.host["pts"][w], o, 5)
        _close(ba.poses[w].cpu().numpy().reshape(-1, 4, 4), Po)
        _close(ba.pts[w].cpu().numpy(), Xo)
    assert torch.equal(ba.pts[3][[0, 17, 299]], ba.pts0[3][[0, 17, 299]])


def test_local_ba_small_window_mixed_batch(ctx):
    """Windows of the MFMA path (k_ba_prepare renumbers their points) side by side in one batch, twice in a row: an ordinary
    one, one whose observations are not grouped by point, one with a point index out of range, one with a repeated (keyframe,
    point) pair, one with unobserved points. The rejected ones are flagged in stats[7] and left as they are -- their slots of
    the renumbered copy are partly unwritten, nothing may index with them -- the others match the CPU solver."""
    import torch
    from trackingbench_slam_amd.ba import BatchedLocalBA
    ba = BatchedLocalBA(ctx, 5, nkf=7, npt=400, iters=5, seed=13, device=torch.device("cuda", 0), distinct=5)
    obs = ba.host["obs"].copy()
    cnt = ba.host["counts"].copy()
    a, b = 40, int(cnt[1]) - 30
    obs[1, [a, b]] = obs[1, [b, a]]                             # not grouped by ascending point any more
    obs["pt"][2, 100] = 5000                                    # out of range
    obs[3, 21] = obs[3, 20]                                     # same (kf, pt) twice (if 20, 21 share the point: a repeat; else ungrouped)
    keep = ~np.isin(obs[4, :cnt[4]]["pt"], (0, 123, 399))
    o4 = obs[4, :cnt[4]][keep]
    obs[4, :len(o4)] = o4
    cnt[4] = len(o4)
    ba.obs.copy_(torch.from_numpy(obs.view(np.uint8).reshape(ba.obs.shape)))
    ba.counts.copy_(torch.from_numpy(cnt))
    for _ in range(2):                                          # the second call finds the first one's tables in the workspace
        ba.run()
        torch.cuda.synchronize()
        st = ba.stats.cpu().numpy()
        assert (st[[1, 2, 3], 7] == -1).all() and st[0, 7] != -1 and st[4, 7] != -1
        for w in (1, 2, 3):
            assert torch.allclose(ba.poses[w], ba.poses0[w], rtol=0, atol=1e-6) and torch.equal(ba.pts[w], ba.pts0[w])
        for w, o in ((0, obs[0, :cnt[0]]), (4, o4)):
            io, Po, Xo, so = oracle.local_ba(K, ba.host["poses"][w], 2, ba.host["pts"][w], o, 5)
            _close(ba.poses[w].cpu().numpy().reshape(-1, 4, 4), Po)
            _close(ba.pts[w].cpu().numpy(), Xo)
        assert torch.equal(ba.pts[4][[0, 123, 399]], ba.pts0[4][[0, 123, 399]])
