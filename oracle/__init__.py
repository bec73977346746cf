"""TEST INFRASTRUCTURE ONLY -- ctypes binding of the CPU oracle (oracle/liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product package (trackingbench_slam_amd) never does.

Parity status: "parity unpinned" against genuine OpenCV 3.3 / g2o / fast_lib -- the
reference holds no golden vectors and those libraries are absent; the
oracle restates their published algorithms next to the reference's own code.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KEYPOINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
MATCH = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])
CORNER = np.dtype([("x", "<i4"), ("y", "<i4"), ("score", "<i4")])
OBS = np.dtype([("u", "<f4"), ("v", "<f4"), ("X", "<f4"), ("Y", "<f4"), ("Z", "<f4"), ("inv_sigma2", "<f4")])
BA_OBS = np.dtype([("kf", "<i4"), ("pt", "<i4"), ("u", "<f4"), ("v", "<f4"), ("inv_sigma2", "<f4")])
CAMERA = np.dtype([("fx", "<f4"), ("fy", "<f4"), ("cx", "<f4"), ("cy", "<f4"), ("width", "<i4"), ("height", "<i4"),
                   ("has_distortion", "<i4"), ("d", "<f4", (5,))])
MAPPOINT = np.dtype([("pos", "<f4", (3,)), ("normal", "<f4", (3,)), ("min_dist", "<f4"), ("max_dist", "<f4"), ("bad", "<i4")])


def build(force=False):
    # ORACLE_LIB picks another build of the same sources (tests/test_oracle_asan.py: liboracle_asan.so under
    # LD_PRELOAD=libasan.so); default is the plain -O2 library.
    name = os.environ.get("ORACLE_LIB", "liboracle.so")
    so = os.path.join(_HERE, name)
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, name], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_ic_angle.restype = C.c_float
        _LIB.orc_shi_tomasi.restype = C.c_float
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u8(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.ndim == 2
    return img


class OracleError(RuntimeError):
    pass


def _chk(rc):
    if rc < 0:
        raise OracleError("oracle error %d" % rc)
    return rc


def scale_factors(n, scale):
    sf = np.zeros(n, np.float32); isf = np.zeros(n, np.float32)
    s2 = np.zeros(n, np.float32); is2 = np.zeros(n, np.float32)
    _chk(lib().orc_scale_factors(n, C.c_float(scale), _p(sf), _p(isf), _p(s2), _p(is2)))
    return sf, isf, s2, is2


def pyramid_sizes(w, h, sf):
    n = len(sf)
    ws = np.zeros(n, np.int32); hs = np.zeros(n, np.int32)
    sf = np.ascontiguousarray(sf, np.float32)
    _chk(lib().orc_pyramid_sizes(w, h, n, _p(sf), _p(ws), _p(hs)))
    return ws, hs


def resize_linear(src, dw, dh):
    src = _u8(src)
    dst = np.zeros((dh, dw), np.uint8)
    _chk(lib().orc_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw))
    return dst


def pyramid(img, nlevels, scale):
    """Frame::ComputePyramid: list of level images (level 0 aliases img)."""
    img = _u8(img)
    sf = scale_factors(nlevels, scale)[0]
    ws, hs = pyramid_sizes(img.shape[1], img.shape[0], sf)
    levels = [img]
    for i in range(1, nlevels):
        levels.append(resize_linear(levels[i - 1], int(ws[i]), int(hs[i])))
    return levels, sf


def fast9(img, th, nms=True):
    img = _u8(img)
    cap = img.size // (4 if nms else 1) + 64
    out = np.zeros(cap, CORNER)
    n = _chk(lib().orc_fast9(_p(img), img.shape[1], img.shape[0], img.strides[0], int(th), int(nms), _p(out), cap))
    return out[:n].copy()


def fast10_nms(img, th=20):
    img = _u8(img)
    cap = img.size // 4 + 64
    out = np.zeros(cap, CORNER)
    n = _chk(lib().orc_fast10_nms(_p(img), img.shape[1], img.shape[0], img.strides[0], int(th), _p(out), cap))
    return out[:n].copy()


def fast_score_map(img, arc=9):
    img = _u8(img)
    out = np.zeros(img.shape, np.int16)
    _chk(lib().orc_fast_score_map(_p(img), img.shape[1], img.shape[0], img.strides[0], arc, _p(out)))
    return out


def gaussian7(img):
    img = _u8(img)
    out = np.zeros_like(img)
    _chk(lib().orc_gaussian7(_p(img), img.shape[1], img.shape[0], img.strides[0], _p(out), out.strides[0]))
    return out


def ic_angle(img, x, y):
    img = _u8(img)
    return float(lib().orc_ic_angle(_p(img), img.strides[0], C.c_float(x), C.c_float(y)))


def orb_descriptor(blurred, x, y, angle):
    img = _u8(blurred)
    d = np.zeros(32, np.uint8)
    lib().orc_orb_descriptor(_p(img), img.strides[0], C.c_float(x), C.c_float(y), C.c_float(angle), _p(d))
    return d


def orb_quotas(sf, target):
    sf = np.ascontiguousarray(sf, np.float32)
    q = np.zeros(len(sf), np.int32)
    _chk(lib().orc_orb_quotas(len(sf), _p(sf), int(target), _p(q)))
    return q


def orb_candidates(img, init_th, min_th):
    img = _u8(img)
    cap = img.size // 4 + 64
    out = np.zeros(cap, CORNER)
    n = _chk(lib().orc_orb_candidates(_p(img), img.shape[1], img.shape[0], img.strides[0],
                                      C.c_float(init_th), C.c_float(min_th), _p(out), cap))
    return out[:n].copy()


def distribute_octtree(cand, min_x, max_x, min_y, max_y, quota, exit_keys=None):
    cand = np.ascontiguousarray(cand, CORNER)
    out = np.zeros(len(cand) + 1, CORNER)
    ek = None if exit_keys is None else np.ascontiguousarray(exit_keys, KEYPOINT)
    n = _chk(lib().orc_distribute_octtree(_p(cand), len(cand), _p(ek), 0 if ek is None else len(ek),
                                          min_x, max_x, min_y, max_y, int(quota), _p(out), len(out)))
    return out[:n].copy()


def _level_args(levels):
    levels = [_u8(l) for l in levels]
    n = len(levels)
    ptrs = (C.c_void_p * n)(*[l.ctypes.data for l in levels])
    ws = np.array([l.shape[1] for l in levels], np.int32)
    hs = np.array([l.shape[0] for l in levels], np.int32)
    st = np.array([l.strides[0] for l in levels], np.int32)
    return levels, ptrs, ws, hs, st


def orb_extract(levels, sf, target, init_th, min_th, exit_keys=None, quotas=None):
    """ORBExtractor::operator() (quotas None) or AddPoints (quotas given). Returns (kps, desc, quotas)."""
    levels, ptrs, ws, hs, st = _level_args(levels)
    sf = np.ascontiguousarray(sf, np.float32)
    cap = sum(int(l.size) for l in levels) // 4 + 64
    kps = np.zeros(cap, KEYPOINT)
    desc = np.zeros((cap, 32), np.uint8)
    use_q = quotas is not None
    q = np.zeros(len(levels), np.int32) if quotas is None else np.ascontiguousarray(quotas, np.int32).copy()
    ek = None if exit_keys is None else np.ascontiguousarray(exit_keys, KEYPOINT)
    n = _chk(lib().orc_orb_extract(ptrs, _p(ws), _p(hs), _p(st), len(levels), _p(sf), int(target),
                                   C.c_float(init_th), C.c_float(min_th), _p(ek), 0 if ek is None else len(ek),
                                   int(use_q), _p(q), _p(kps), _p(desc), cap))
    return kps[:n].copy(), desc[:n].copy(), q


def fastgrid_extract(levels, inv_sf, target, threshold, occupancy=None):
    levels, ptrs, ws, hs, st = _level_args(levels)
    inv_sf = np.ascontiguousarray(inv_sf, np.float32)
    cap = max(int(target), 1) * 4 + 4096
    kps = np.zeros(cap, KEYPOINT)
    occ = None if occupancy is None else np.ascontiguousarray(occupancy, np.uint8)
    n = _chk(lib().orc_fastgrid_extract(ptrs, _p(ws), _p(hs), _p(st), len(levels), _p(inv_sf), int(target),
                                        C.c_float(threshold), _p(occ), 0 if occ is None else len(occ), _p(kps), cap))
    return kps[:n].copy()


def shi_tomasi(img, u, v):
    img = _u8(img)
    return float(lib().orc_shi_tomasi(_p(img), img.shape[1], img.shape[0], img.strides[0], int(u), int(v)))


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return int(lib().orc_descriptor_distance(_p(a), _p(b)))


def three_maxima(sizes):
    sizes = np.ascontiguousarray(sizes, np.int32)
    i1, i2, i3 = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    lib().orc_three_maxima(_p(sizes), len(sizes), C.byref(i1), C.byref(i2), C.byref(i3))
    return i1.value, i2.value, i3.value


def _desc(d):
    d = np.ascontiguousarray(d, np.uint8)
    if d.size == 0:
        d = d.reshape(0, 32)
    assert d.ndim == 2 and d.shape[1] == 32
    return d


def bf_match(d1, d2, crosscheck=True):
    d1, d2 = _desc(d1), _desc(d2)
    out = np.zeros(max(len(d1), 1), MATCH)
    n = _chk(lib().orc_bf_match(_p(d1), len(d1), _p(d2), len(d2), int(crosscheck), _p(out), len(out)))
    return out[:n].copy()


def search_by_bf(d1, d2, ratio, min_th):
    d1, d2 = _desc(d1), _desc(d2)
    out = np.zeros(max(len(d1), 1), MATCH)
    n = _chk(lib().orc_search_by_bf(_p(d1), len(d1), _p(d2), len(d2), C.c_float(ratio), C.c_float(min_th),
                                    _p(out), len(out)))
    return out[:n].copy()


def search_by_violence(k1, d1, k2, d2, img2_w, img2_h, min_level=0, max_level=1, radius=10.0,
                       th_low=50, nratio=0.0, histo_len=30, check_orientation=True):
    k1 = np.ascontiguousarray(k1, KEYPOINT); k2 = np.ascontiguousarray(k2, KEYPOINT)
    d1, d2 = _desc(d1), _desc(d2)
    out = np.zeros(max(len(k1), 1), MATCH)
    n = _chk(lib().orc_search_by_violence(_p(k1), _p(d1), len(k1), _p(k2), _p(d2), len(k2), int(img2_w), int(img2_h),
                                          int(min_level), int(max_level), C.c_float(radius), int(th_low),
                                          C.c_float(nratio), int(histo_len), int(check_orientation), _p(out), len(out)))
    return out[:n].copy()


def _fv(fv):
    """DBoW2 feature vector {node id: [feature indices]} -> (nodes u32 ascending, start i32, items u32)."""
    nodes = np.array(sorted(fv), np.uint32)
    start = np.zeros(len(nodes) + 1, np.int32)
    items = []
    for i, nd in enumerate(nodes):
        items.extend(int(x) for x in fv[int(nd)])
        start[i + 1] = len(items)
    return nodes, start, np.array(items, np.uint32)


def search_by_bow(k1, d1, fv1, k2, d2, fv2, has_mp2=None, map_point_only=False, th_low=50, nratio=0.0, histo_len=30,
                  check_orientation=True):
    """Matcher::searchByBow(F1, F2, MapPointOnly), matcher.cpp:619-721; fv1 / fv2: the frames' DBoW2 feature vectors as
    dicts {node id: [feature indices]}."""
    k1 = np.ascontiguousarray(k1, KEYPOINT); k2 = np.ascontiguousarray(k2, KEYPOINT)
    d1, d2 = _desc(d1), _desc(d2)
    n1a, s1a, i1a = _fv(fv1)
    n2a, s2a, i2a = _fv(fv2)
    hm = None if has_mp2 is None else np.ascontiguousarray(has_mp2, np.uint8)
    out = np.zeros(max(len(i1a), 1), MATCH)
    n = _chk(lib().orc_search_by_bow(_p(k1), _p(d1), len(k1), _p(n1a), _p(s1a), _p(i1a), len(n1a), _p(k2), _p(d2), len(k2), _p(hm),
                                     _p(n2a), _p(s2a), _p(i2a), len(n2a), int(map_point_only), int(th_low), C.c_float(nratio),
                                     int(histo_len), int(check_orientation), _p(out), len(out)))
    return out[:n].copy()


def camera(fx, fy, cx, cy, width, height, dist=None):
    """tb_camera record (PinholeCamera of the reference); dist = (k1, k2, p1, p2, k3) or None."""
    c = np.zeros(1, CAMERA)
    c["fx"], c["fy"], c["cx"], c["cy"], c["width"], c["height"] = fx, fy, cx, cy, width, height
    if dist is not None:
        c["has_distortion"] = 1
        c["d"][0] = np.asarray(dist, np.float32)
    return c


def search_by_projection(Tcw1, cam1, img1_w, img1_h, k1, d1, taken1, k2, mp2, mp2_desc, scale_factors, nratio,
                         th_high=100, histo_len=30, check_orientation=True):
    """Matcher::searchByProjection(F1, F2) (matcher.cpp:406-531); mp2 / mp2_desc aligned with F2's keys."""
    Tcw1 = np.ascontiguousarray(Tcw1, np.float32).reshape(16)
    cam1 = np.ascontiguousarray(cam1, CAMERA)
    k1 = np.ascontiguousarray(k1, KEYPOINT); k2 = np.ascontiguousarray(k2, KEYPOINT)
    d1, mp2_desc = _desc(d1), _desc(mp2_desc)
    taken1 = np.ascontiguousarray(taken1, np.uint8)
    mp2 = np.ascontiguousarray(mp2, MAPPOINT)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    out = np.zeros(max(len(k2), 1), MATCH)
    n = _chk(lib().orc_search_by_projection(_p(Tcw1), _p(cam1), int(img1_w), int(img1_h), _p(k1), _p(d1), _p(taken1), len(k1),
                                            _p(k2), _p(mp2), _p(mp2_desc), len(k2), _p(sf), len(sf), C.c_float(nratio),
                                            int(th_high), int(histo_len), int(check_orientation), _p(out), len(out)))
    return out[:n].copy()


def search_by_projection_map(Tcw1, cam1, img1_w, img1_h, k1, d1, taken1, mps, mp_desc, scale_factors, nratio, radio,
                             th_high=100):
    """Matcher::searchByProjection(map, F1, radio) (matcher.cpp:539-617)."""
    Tcw1 = np.ascontiguousarray(Tcw1, np.float32).reshape(16)
    cam1 = np.ascontiguousarray(cam1, CAMERA)
    k1 = np.ascontiguousarray(k1, KEYPOINT)
    d1, mp_desc = _desc(d1), _desc(mp_desc)
    taken1 = np.ascontiguousarray(taken1, np.uint8)
    mps = np.ascontiguousarray(mps, MAPPOINT)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    out = np.zeros(max(len(mps), 1), MATCH)
    n = _chk(lib().orc_search_by_projection_map(_p(Tcw1), _p(cam1), int(img1_w), int(img1_h), _p(k1), _p(d1), _p(taken1),
                                                len(k1), _p(mps), _p(mp_desc), len(mps), _p(sf), len(sf),
                                                C.c_float(nratio), C.c_float(radio), int(th_high), _p(out), len(out)))
    return out[:n].copy()


def bow_transform(voc, desc, levelsup=4):
    """voc->transform(features, BowVector, FeatureVector, levelsup) per feature (Frame::SetBow, Frame.cpp:267-270;
    TemplatedVocabulary.h:1124-1260): (word_ids i32, weights f64, node_ids i32). voc: synth.Vocabulary."""
    d = _desc(desc)
    n = len(d)
    wid = np.zeros(max(n, 1), np.int32); wt = np.zeros(max(n, 1), np.float64); nid = np.zeros(max(n, 1), np.int32)
    _chk(lib().orc_bow_transform(C.byref(voc.c), _p(d), n, int(levelsup), _p(wid), _p(wt), _p(nid)))
    return wid[:n], wt[:n], nid[:n]


def bow_containers(word_ids, weights, node_ids, weighting=0, scoring=0):
    """The two containers of TemplatedVocabulary::transform(features, v, fv, levelsup) (TemplatedVocabulary.h:1124-1188):
    BowVector {word: value} (TF / TF-IDF add the weights in feature order, IDF / BINARY keep the first; then the scoring
    object's normalisation, BowVector.cpp:57-80) and FeatureVector {node: [feature indices]}; stopped words (weight 0) enter neither."""
    bv, fv = {}, {}
    for i, (w, wt, n) in enumerate(zip(word_ids.tolist(), weights.tolist(), node_ids.tolist())):
        if wt > 0:
            if weighting in (0, 1):
                bv[w] = bv.get(w, 0.0) + wt
            else:
                bv.setdefault(w, wt)
            fv.setdefault(n, []).append(i)
    must, l2 = scoring != 5, scoring == 1
    if bv and not must and weighting in (0, 1):
        nd = float(len(bv))
        bv = {k: v / nd for k, v in bv.items()}
    if must:
        norm = 0.0
        for k in sorted(bv):
            norm += bv[k] * bv[k] if l2 else abs(bv[k])
        if l2:
            norm = float(np.sqrt(norm))
        if norm > 0:
            bv = {k: v / norm for k, v in bv.items()}
    return dict(sorted(bv.items())), dict(sorted(fv.items()))


def stereo_tracks_to_obs(kl, kr, matches, K, bf, inv_sigma2):
    """The composition of the timed configuration: left <-> right matches -> PoseOptimization rows (stereo depth of the left key,
    LocalBA.cpp:60-64, observed at the right key's pixel, LocalBA.cpp:333-363)."""
    kl = np.ascontiguousarray(kl, KEYPOINT); kr = np.ascontiguousarray(kr, KEYPOINT)
    m = np.ascontiguousarray(matches, MATCH)
    Kf = np.ascontiguousarray(K, np.float32)
    sig = np.ascontiguousarray(inv_sigma2, np.float32)
    out = np.zeros(max(len(m), 1), OBS)
    n = _chk(lib().orc_stereo_tracks_to_obs(_p(kl), _p(kr), _p(m), len(m), _p(Kf), C.c_float(bf), _p(sig), len(sig), _p(out), len(out)))
    return out[:n].copy()


def pose_opt(K, Tcw, obs, outlier=None):
    K = np.ascontiguousarray(K, np.float64)
    Tcw = np.ascontiguousarray(Tcw, np.float32).reshape(16)
    obs = np.ascontiguousarray(obs, OBS)
    outl = np.zeros(len(obs), np.uint8) if outlier is None else np.ascontiguousarray(outlier, np.uint8).copy()
    out = np.zeros(16, np.float32)
    stats = np.zeros(8, np.float64)
    n = _chk(lib().orc_pose_opt(_p(K), _p(Tcw), _p(obs), len(obs), _p(outl), _p(out), _p(stats)))
    return n, out.reshape(4, 4), outl, stats


def local_ba(K, poses, nfixed, pts, obs, iters=10):
    K = np.ascontiguousarray(K, np.float64)
    poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 16).copy()
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3).copy()
    obs = np.ascontiguousarray(obs, BA_OBS)
    stats = np.zeros(8, np.float64)
    n = _chk(lib().orc_local_ba(_p(K), len(poses), int(nfixed), _p(poses), len(pts), _p(pts), _p(obs), len(obs),
                                int(iters), _p(stats)))
    return n, poses.reshape(-1, 4, 4), pts, stats


def pyr_down(img):
    """cv::pyrDown (one level of cv::buildOpticalFlowPyramid)."""
    img = _u8(img)
    h, w = img.shape
    out = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
    _chk(lib().orc_pyr_down(_p(img), w, h, w, _p(out), out.shape[1]))
    return out


def optical_flow_pyr_lk(prev, nxt, prev_pts, win=21, max_level=3):
    """cv::calcOpticalFlowPyrLK(prev, next, prev_pts, ..., Size(win, win), max_level), default criteria / flags
    (restated from OpenCV 3.3, parity unpinned). Returns (next_pts [n,2], status [n] u8, err [n], top level used)."""
    prev, nxt = _u8(prev), _u8(nxt)
    assert prev.shape == nxt.shape
    h, w = prev.shape
    pts = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2)
    n = len(pts)
    out = np.zeros((max(n, 1), 2), np.float32)
    status = np.zeros(max(n, 1), np.uint8)
    err = np.zeros(max(n, 1), np.float32)
    top = _chk(lib().orc_optical_flow_pyr_lk(_p(prev), _p(nxt), w, h, w, _p(pts), n, int(win), int(max_level), _p(out),
                                             _p(status), _p(err)))
    return out[:n], status[:n], err[:n], top


def search_by_opflow(img1, img2, cam1, keys2_xy, equalized=False, reject=False):
    """Matcher::searchByOPFlow(F1, F2, cur_points, equalized, reject), matcher.cpp:724-768.
    Returns (cur_points [n,2], matched indices i (queryIdx = trainIdx = i))."""
    img1, img2 = _u8(img1), _u8(img2)
    h, w = img1.shape
    pts = np.ascontiguousarray(keys2_xy, np.float32).reshape(-1, 2)
    n = len(pts)
    cur = np.zeros((max(n, 1), 2), np.float32)
    idx = np.zeros(max(n, 1), np.int32)
    m = _chk(lib().orc_search_by_opflow(_p(img1), _p(img2), w, h, w, _p(cam1), _p(pts), n, int(equalized), int(reject), _p(cur),
                                        _p(idx)))
    return cur[:n], idx[:m].copy()


def find_fundamental_ransac(pts1, pts2, thresh=1.0, conf=0.99):
    """cv::findFundamentalMat(pts1, pts2, FM_RANSAC, thresh, conf, mask) as Matcher::rejectWithF calls it (matcher.cpp:872),
    restated (parity unpinned). Returns (ok, mask [n] u8, F [3,3] f64, RANSAC iterations run); ok = 0: no mask comes back."""
    p1 = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(pts2, np.float32).reshape(-1, 2)
    assert len(p1) == len(p2)
    n = len(p1)
    mask = np.zeros(max(n, 1), np.uint8)
    F = np.zeros(9, np.float64)
    it = C.c_int(0)
    rc = lib().orc_find_fundamental_ransac(_p(p1), _p(p2), n, C.c_double(thresh), C.c_double(conf), _p(mask), _p(F), C.byref(it))
    if rc < 0:
        raise OracleError("find_fundamental_ransac: error %d" % rc)
    return rc, mask[:n], F.reshape(3, 3), it.value


def reject_with_f(cur_pts, last_pts, status):
    """Matcher::rejectWithF(cur_pts, last_pts, status), matcher.cpp:853-881. Returns the updated status flags."""
    cur = np.ascontiguousarray(cur_pts, np.float32).reshape(-1, 2)
    last = np.ascontiguousarray(last_pts, np.float32).reshape(-1, 2)
    st = np.ascontiguousarray(status, np.uint8).copy()
    assert len(cur) == len(last) == len(st)
    rc = lib().orc_reject_with_f(_p(cur), _p(last), len(st), _p(st))
    if rc < 0:
        raise OracleError("reject_with_f: error %d" % rc)
    return st


def add_map_points_by_stereo(img_stereo, img_current, cam_stereo, keys_xy, bf):
    """LocalBA::AddMapPointsByStereo(current_frame, stereo_frame, bf, fx), LocalBA.cpp:46-68: depth per key of the current
    frame (-1 where the tracker, the frame test or the RANSAC stage dropped it)."""
    a, b = _u8(img_stereo), _u8(img_current)
    h, w = a.shape
    pts = np.ascontiguousarray(keys_xy, np.float32).reshape(-1, 2)
    n = len(pts)
    depth = np.zeros(max(n, 1), np.float32)
    m = lib().orc_add_map_points_by_stereo(_p(a), _p(b), w, h, w, _p(cam_stereo), _p(pts), n, C.c_float(bf), _p(depth))
    if m < 0:
        raise OracleError("add_map_points_by_stereo: %d" % m)
    return depth[:n]


def clahe(img, clip_limit=3.0, tiles=(8, 8)):
    """Frame::Equalize (Frame.cpp:453-458): cv::createCLAHE(3.0, Size(8, 8))->apply, restated (parity unpinned)."""
    img = _u8(img)
    h, w = img.shape
    out = np.zeros_like(img)
    _chk(lib().orc_clahe(_p(img), w, h, w, C.c_double(clip_limit), int(tiles[0]), int(tiles[1]), _p(out), w))
    return out
