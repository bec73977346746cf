"""CPU tests of the optical-flow oracle (oracle/oracle_flow.cpp; SURVEY 8f row 2, first part). OpenCV is not
available and the reference holds no vectors for searchByOPFlow: PARITY UNPINNED -- these check the restated
algorithm against properties it must have (identity, known translation, bounds handling)."""
import numpy as np

import oracle
from trackingbench_slam_amd import synth


def _keys(img, n=300):
    lv, sf = oracle.pyramid(img, 8, 0.8)
    k, _, _ = oracle.orb_extract(lv, sf, n, 80, 30)
    return np.stack([k["x"], k["y"]], 1).astype(np.float32)


def test_pyr_down_constant_and_size():
    img = np.full((37, 51), 93, np.uint8)
    d = oracle.pyr_down(img)
    assert d.shape == (19, 26) and (d == 93).all()
    g = synth.frame(1, 64, 48)
    d = oracle.pyr_down(g)
    # interior pixel against the 5x5 binomial kernel
    k = np.array([1, 4, 6, 4, 1])
    assert d[5, 7] == (int((np.outer(k, k) * g[8:13, 12:17].astype(int)).sum()) + 128) >> 8


def test_identity_and_translation():
    L = synth.frame(3, 640, 480)
    pts = _keys(L)
    nxt, st, err, top = oracle.optical_flow_pyr_lk(L, L, pts)
    assert top == 3 and st.all() and np.abs(nxt - pts).max() < 1e-3 and err.max() < 0.01
    S = np.roll(L, (3, 5), (0, 1))
    nxt, st, err, _ = oracle.optical_flow_pyr_lk(L, S, pts)
    inner = (pts[:, 0] > 40) & (pts[:, 0] < 600) & (pts[:, 1] > 40) & (pts[:, 1] < 440) & (st > 0)
    assert inner.sum() > 100 and np.percentile(np.abs((nxt - pts)[inner] - [5, 3]).max(1), 95) < 0.02


def test_small_image_and_outside_points():
    img = synth.frame(2, 96, 64)           # levels: 96x64, 48x32, 24x16 (stop: 16 <= 21) -> top level 1
    pts = np.array([[10, 10], [-50, 5], [95, 63], [500, 500]], np.float32)
    nxt, st, err, top = oracle.optical_flow_pyr_lk(img, img, pts)
    assert top == 1
    assert st[1] == 0 and st[3] == 0       # window entirely outside the image
    assert len(oracle.optical_flow_pyr_lk(img, img, np.zeros((0, 2), np.float32))[0]) == 0


def test_search_by_opflow_filters_frame():
    L, R = synth.frame(4, 640, 480, stereo=True)
    pts = _keys(L)
    cam = oracle.camera(500, 500, 320, 240, 640, 480)
    cur, idx = oracle.search_by_opflow(R, L, cam, pts)
    nxt, st, _, _ = oracle.optical_flow_pyr_lk(L, R, pts)
    assert np.array_equal(cur, nxt)
    u, v = cur[:, 0].astype(np.int32), cur[:, 1].astype(np.int32)
    keep = (st > 0) & (u >= 0) & (u < 640) & (v >= 0) & (v < 480)
    assert np.array_equal(idx, np.nonzero(keep)[0]) and len(idx) > 50


def test_clahe_properties():
    g = synth.frame(1, 640, 480)
    low = (g // 4 + 90).astype(np.uint8)
    e = oracle.clahe(low)
    assert e.std() > 1.5 * low.std()                      # contrast is stretched
    assert np.array_equal(oracle.clahe(low), e)           # deterministic
    # without a clip limit a single tile is plain histogram equalisation: LUT = round(cdf * 255 / area)
    one = oracle.clahe(low, 0.0, (1, 1))
    hist = np.bincount(low.ravel(), minlength=256)
    lut = np.clip(np.rint(np.cumsum(hist).astype(np.float32) * np.float32(255.0 / low.size)), 0, 255).astype(np.uint8)
    assert np.array_equal(one, lut[low])
    # sizes that are not multiples of the tile grid take the padded-histogram path
    odd = synth.frame(2, 333, 211)
    assert oracle.clahe(odd).shape == odd.shape
