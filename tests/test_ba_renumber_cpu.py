"""CPU check of the idea behind k_ba_prepare (no GPU): a window whose points are renumbered in visibility-pattern order
(observations regrouped, each point's edges in their old order) is the same least-squares problem -- the CPU solver reaches the
same poses, and the points of the renumbered window are the permuted points of the original one. Sums over points run in a
different order, so the comparison is to rounding, not bit for bit."""
import numpy as np

import oracle
from trackingbench_slam_amd import synth
from trackingbench_slam_amd.ba import _pattern_sorted

K = (718.856, 718.856, 607.1928, 185.2157)


def test_pattern_sorted_window_is_the_same_problem():
    nfixed = 2
    prob = synth.ba_problem(5, 7, 400, K, obs_per_pt=4)
    Pt, Pi, Xt, Xi, obs = prob
    Pt2, Pi2, Xt2, Xi2, obs2 = _pattern_sorted(prob, nfixed)
    # the renumbering: ascending visibility mask, observations still grouped by ascending point
    free = obs2["kf"] >= nfixed
    mask = np.zeros(len(Xi2), np.int64)
    np.bitwise_or.at(mask, obs2["pt"][free], np.int64(1) << (obs2["kf"][free] - nfixed).astype(np.int64))
    assert (np.diff(mask) >= 0).all() and (np.diff(obs2["pt"]) >= 0).all() and len(obs2) == len(obs)
    assert sorted(map(tuple, Xi2.tolist())) == sorted(map(tuple, Xi.tolist()))
    i1, P1, X1, s1 = oracle.local_ba(K, Pi, nfixed, Xi, obs, 8)
    i2, P2, X2, s2 = oracle.local_ba(K, Pi2, nfixed, Xi2, obs2, 8)
    assert i1 == i2 and np.isclose(s1[2], s2[2], rtol=1e-9)
    assert np.allclose(P1, P2, rtol=0, atol=1e-6)
    # X2[r] is the optimised position of the point that was perm[r]: match them through the initial positions
    order1 = np.lexsort(Xi.T[::-1]); order2 = np.lexsort(Xi2.T[::-1])
    assert np.allclose(X1[order1], X2[order2], rtol=0, atol=1e-5)
