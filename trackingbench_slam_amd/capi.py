"""ctypes binding of the C ABI (include/tb_capi.h) exported by libtb_hip.so.

This is plumbing, not the product: every call goes straight into the HIP library.  There is no
CPU fallback -- if the library is missing or no GPU is usable the calls raise.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TB_HIP_LIB") or os.path.join(_HERE, "libtb_hip.so")   # TB_HIP_LIB: experiment builds
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

TB_OK, TB_EINVAL, TB_ENOMEM, TB_ECAPACITY, TB_EUNSUPPORTED, TB_EDEVICE, TB_ESTATE = 0, -1, -2, -3, -4, -5, -6

# every symbol include/tb_capi.h declares (checked by tests/test_capi_exports.py)
EXPORTS = [
    "tb_create", "tb_destroy", "tb_last_error", "tb_strerror", "tb_version", "tb_set_stream", "tb_synchronize",
    "tb_profile_enable", "tb_profile_only", "tb_profile_report", "tb_debug_force_dense_fast", "tb_measure_copy_seconds", "tb_set_concurrency", "tb_pack_rows_dev",
    "tb_scale_factors", "tb_pyramid_sizes", "tb_orb_quotas",
    "tb_extractor_create", "tb_extractor_destroy", "tb_extractor_set_images_host", "tb_extractor_set_images_dev",
    "tb_extractor_set_levels_host", "tb_extractor_build_pyramid", "tb_extractor_get_level_host", "tb_extractor_orb",
    "tb_extractor_fastgrid", "tb_extractor_counts_host", "tb_extractor_results_host", "tb_extractor_results_dev",
    "tb_extractor_candidates_host", "tb_extractor_copy_results_dev",
    "tb_pyramid", "tb_fast_detect", "tb_orb_extract", "tb_fastgrid_extract",
    "tb_descriptor_distance", "tb_three_maxima", "tb_match_bf", "tb_search_by_bf", "tb_search_by_bf_batch_dev",
    "tb_search_by_violence", "tb_search_by_bow", "tb_search_by_projection", "tb_search_by_projection_map", "tb_frame_grid_batch_dev",
    "tb_search_by_projection_batch_dev", "tb_search_by_projection_map_batch_dev",
    "tb_search_by_violence_batch_dev", "tb_stereo_tracks_to_obs_batch_dev", "tb_vocab_create", "tb_vocab_destroy", "tb_bow_transform",
    "tb_bow_transform_batch_dev", "tb_search_by_bow_batch_dev", "tb_pose_opt", "tb_pose_opt_batch_dev", "tb_local_ba", "tb_local_ba_batch_dev",
    "tb_clahe", "tb_clahe_dev", "tb_optical_flow_pyr_lk", "tb_optical_flow_pyr_lk_dev", "tb_optical_flow_pyr_lk_batch_dev", "tb_search_by_opflow", "tb_search_by_opflow_batch_dev",
    "tb_find_fundamental_ransac", "tb_reject_with_f", "tb_reject_with_f_batch_dev", "tb_add_map_points_by_stereo", "tb_add_map_points_by_stereo_batch_dev",
    "tb_batch_run",
]


class TBError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("tb error %d: %s" % (code, msg))
        self.code = code


def build(force=False):
    """Compile libtb_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-j8"], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise TBError(TB_EDEVICE, "libtb_hip.so is not built (run __graft_entry__.build())")
        try:
            # One HIP runtime per process: when torch is installed its bundled libamdhip64 must be the one that
            # gets loaded (loading ROCm's copy first makes torch's later initialisation find no GPU).
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.tb_last_error.restype = C.c_char_p
        L.tb_strerror.restype = C.c_char_p
        L.tb_version.restype = C.c_char_p
        L.tb_destroy.restype = None
        L.tb_extractor_destroy.restype = None
        L.tb_three_maxima.restype = None
        L.tb_vocab_destroy.restype = None
        L.tb_vocab_destroy.argtypes = [C.c_void_p]
        L.tb_last_error.argtypes = [C.c_void_p]
        L.tb_destroy.argtypes = [C.c_void_p]
        L.tb_extractor_destroy.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u8img(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.ndim == 2
    return img


def scale_factors(n, scale):
    sf = np.zeros(n, np.float32); isf = np.zeros(n, np.float32)
    s2 = np.zeros(n, np.float32); is2 = np.zeros(n, np.float32)
    rc = lib().tb_scale_factors(n, C.c_float(scale), _p(sf), _p(isf), _p(s2), _p(is2))
    if rc:
        raise TBError(rc, "tb_scale_factors")
    return sf, isf, s2, is2


def pyramid_sizes(w, h, sf):
    sf = np.ascontiguousarray(sf, np.float32)
    ws = np.zeros(len(sf), np.int32); hs = np.zeros(len(sf), np.int32)
    rc = lib().tb_pyramid_sizes(int(w), int(h), len(sf), _p(sf), _p(ws), _p(hs))
    if rc:
        raise TBError(rc, "tb_pyramid_sizes")
    return ws, hs


def orb_quotas(sf, target):
    sf = np.ascontiguousarray(sf, np.float32)
    q = np.zeros(len(sf), np.int32)
    rc = lib().tb_orb_quotas(len(sf), _p(sf), int(target), _p(q))
    if rc:
        raise TBError(rc, "tb_orb_quotas")
    return q


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return int(lib().tb_descriptor_distance(_p(a), _p(b)))


def three_maxima(sizes):
    sizes = np.ascontiguousarray(sizes, np.int32)
    i1, i2, i3 = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    lib().tb_three_maxima(_p(sizes), len(sizes), C.byref(i1), C.byref(i2), C.byref(i3))
    return i1.value, i2.value, i3.value


class BatchParams(C.Structure):
    """tb_batch_params of include/tb_capi.h"""
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("nlevels", C.c_int), ("scale", C.c_float), ("target", C.c_int),
                ("init_th", C.c_float), ("min_th", C.c_float), ("bf_ratio", C.c_float), ("bf_min_th", C.c_float)]


def batch_run(contexts, left, right, nlevels=8, scale=0.8, target=2000, init_th=80.0, min_th=30.0, bf_ratio=10.0,
              bf_min_th=30.0, cap=None):
    """tb_batch_run: a batch of stereo frames (uint8 [F, H, W] each side) sharded over `contexts` (one per GPU, or several on
    one), pyramid -> ORB (both sides) -> searchByBF left <-> right per shard, records gathered on the host.
    Returns per frame (kps_left, desc_left, kps_right, desc_right, matches)."""
    left = np.ascontiguousarray(left, np.uint8); right = np.ascontiguousarray(right, np.uint8)
    F, H, W = left.shape
    assert right.shape == left.shape and len(contexts) >= 1
    cap = int(cap or (target + 512))
    prm = BatchParams(W, H, int(nlevels), float(scale), int(target), float(init_th), float(min_th), float(bf_ratio), float(bf_min_th))
    kps = np.zeros((2, F, cap), KEYPOINT); desc = np.zeros((2, F, cap, 32), np.uint8); cnt = np.zeros((2, F), np.int32)
    mt = np.zeros((F, cap), MATCH); mc = np.zeros(F, np.int32)
    hs = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    rc = lib().tb_batch_run(hs, len(contexts), C.byref(prm), F, _p(left), _p(right), W, C.c_size_t(W * H), cap, _p(kps), _p(desc),
                            _p(cnt), _p(mt), _p(mc))
    if rc != 0:
        msgs = [lib().tb_last_error(c._h).decode() for c in contexts]
        raise TBError(rc, "; ".join(m for m in msgs if m) or lib().tb_strerror(rc).decode())
    return [(kps[0, f, :cnt[0, f]].copy(), desc[0, f, :cnt[0, f]].copy(), kps[1, f, :cnt[1, f]].copy(), desc[1, f, :cnt[1, f]].copy(),
             mt[f, :mc[f]].copy()) for f in range(F)]


class Context:
    """tb_ctx: one GPU + one HIP stream."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        rc = lib().tb_create(int(device), C.byref(self._h))
        if rc:
            raise TBError(rc, "tb_create(device=%d): %s" % (device, lib().tb_strerror(rc).decode()))
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if self._h:
            lib().tb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc:
            raise TBError(rc, lib().tb_last_error(self._h).decode() or lib().tb_strerror(rc).decode())

    def set_stream(self, stream_ptr):
        """Run the context on an existing HIP stream. None = the context's own (non-blocking) stream. Handle 0 -- HIP's
        legacy null stream, which is what torch's DEFAULT stream reports -- is refused: the C ABI reads NULL as "own
        stream", so the caller would believe it is ordered against torch's default stream when it is not. Pass a
        torch.cuda.Stream() handle and issue the torch-side work under `with torch.cuda.stream(...)`."""
        if stream_ptr is None:
            self.check(lib().tb_set_stream(self._h, None))
            return
        if int(stream_ptr) == 0:
            raise ValueError("stream handle 0 (the legacy null stream / torch's default stream) is not accepted: "
                             "create a torch.cuda.Stream() and pass its .cuda_stream, or pass None")
        self.check(lib().tb_set_stream(self._h, C.c_void_p(int(stream_ptr))))

    def synchronize(self):
        self.check(lib().tb_synchronize(self._h))

    def profile_enable(self, on=True, only=None):
        """Per-kernel HIP-event timing on the context's stream; `only`: time just that kernel (None: all)."""
        self.check(lib().tb_profile_only(self._h, only.encode() if only else None))
        self.check(lib().tb_profile_enable(self._h, int(on)))

    def measure_copy_seconds(self, src_ptr, dst_ptr, nbytes, reps=10):
        """Average seconds per device-to-device copy of nbytes with the library's 16-byte-per-lane kernel."""
        sec = C.c_double(0)
        self.check(lib().tb_measure_copy_seconds(self._h, C.c_void_p(src_ptr), C.c_void_p(dst_ptr), C.c_size_t(nbytes), int(reps), C.byref(sec)))
        return sec.value

    def set_concurrency(self, peers):
        """tb_set_concurrency: `peers` contexts share this GPU at the same time (launch-shape hint)."""
        self.check(lib().tb_set_concurrency(self._h, int(peers)))

    def force_dense_fast(self, on=True):
        """Test hook: every FAST block takes the any-density path (same results)."""
        self.check(lib().tb_debug_force_dense_fast(self._h, int(on)))

    def profile_report(self):
        """{kernel name: (calls, total_ms)} accumulated since profile_enable(True)."""
        buf = C.create_string_buffer(8192)
        self.check(lib().tb_profile_report(self._h, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, calls, ms = line.split()
            out[name] = (int(calls), float(ms))
        return out

    # ---- single-frame operator forms
    def pyramid(self, img, nlevels, scale):
        img = _u8img(img)
        sf = scale_factors(nlevels, scale)[0]
        ws, hs = pyramid_sizes(img.shape[1], img.shape[0], sf)
        levels = [img] + [np.zeros((int(hs[i]), int(ws[i])), np.uint8) for i in range(1, nlevels)]
        ptrs = (C.c_void_p * nlevels)(*[l.ctypes.data for l in levels])
        st = np.array([l.strides[0] for l in levels], np.int32)
        self.check(lib().tb_pyramid(self._h, _p(img), img.shape[1], img.shape[0], img.strides[0], nlevels, _p(sf), ptrs, _p(st)))
        return levels, sf

    def fast_detect(self, img, th, nms=True):
        img = _u8img(img)
        cap = img.size // (4 if nms else 1) + 4096
        out = np.zeros(cap, CORNER)
        n = C.c_int(0)
        self.check(lib().tb_fast_detect(self._h, _p(img), img.shape[1], img.shape[0], img.strides[0], int(th), int(nms),
                                        _p(out), cap, C.byref(n)))
        return out[:n.value].copy()

    @staticmethod
    def _level_args(levels):
        levels = [_u8img(l) for l in levels]
        n = len(levels)
        ptrs = (C.c_void_p * n)(*[l.ctypes.data for l in levels])
        ws = np.array([l.shape[1] for l in levels], np.int32)
        hs = np.array([l.shape[0] for l in levels], np.int32)
        st = np.array([l.strides[0] for l in levels], np.int32)
        return levels, ptrs, ws, hs, st

    def orb_extract(self, levels, sf, target, init_th, min_th, exit_keys=None, quotas=None):
        levels, ptrs, ws, hs, st = self._level_args(levels)
        sf = np.ascontiguousarray(sf, np.float32)
        cap = int(target) + 64 * len(levels) + 64
        if quotas is not None:
            cap = max(cap, int(np.sum(quotas)) + 64 * len(levels) + 64)
        kps = np.zeros(cap, KEYPOINT)
        desc = np.zeros((cap, 32), np.uint8)
        q = np.zeros(len(levels), np.int32) if quotas is None else np.ascontiguousarray(quotas, np.int32).copy()
        ek = None if exit_keys is None else np.ascontiguousarray(exit_keys, KEYPOINT)
        n = C.c_int(0)
        self.check(lib().tb_orb_extract(self._h, ptrs, _p(ws), _p(hs), _p(st), len(levels), _p(sf), int(target),
                                        C.c_float(init_th), C.c_float(min_th), _p(ek), 0 if ek is None else len(ek),
                                        int(quotas is not None), _p(q), _p(kps), _p(desc), cap, C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy(), q

    def fastgrid_extract(self, levels, inv_sf, target, threshold, occupancy=None):
        levels, ptrs, ws, hs, st = self._level_args(levels)
        inv_sf = np.ascontiguousarray(inv_sf, np.float32)
        cap = int(target) * 2 + 4096
        kps = np.zeros(cap, KEYPOINT)
        occ = None if occupancy is None else np.ascontiguousarray(occupancy, np.uint8)
        n = C.c_int(0)
        self.check(lib().tb_fastgrid_extract(self._h, ptrs, _p(ws), _p(hs), _p(st), len(levels), _p(inv_sf), int(target),
                                             C.c_float(threshold), _p(occ), 0 if occ is None else len(occ), _p(kps), cap,
                                             C.byref(n)))
        return kps[:n.value].copy()

    @staticmethod
    def _desc(d):
        d = np.ascontiguousarray(d, np.uint8)
        if d.size == 0:
            d = d.reshape(0, 32)
        assert d.ndim == 2 and d.shape[1] == 32
        return d

    def bf_match(self, d1, d2, crosscheck=True):
        d1, d2 = self._desc(d1), self._desc(d2)
        out = np.zeros(max(len(d1), 1), MATCH)
        n = C.c_int(0)
        self.check(lib().tb_match_bf(self._h, _p(d1), len(d1), _p(d2), len(d2), int(crosscheck), _p(out), len(out), C.byref(n)))
        return out[:n.value].copy()

    def search_by_bf(self, d1, d2, ratio, min_th):
        d1, d2 = self._desc(d1), self._desc(d2)
        out = np.zeros(max(len(d1), 1), MATCH)
        n = C.c_int(0)
        self.check(lib().tb_search_by_bf(self._h, _p(d1), len(d1), _p(d2), len(d2), C.c_float(ratio), C.c_float(min_th),
                                         _p(out), len(out), C.byref(n)))
        return out[:n.value].copy()

    def search_by_violence(self, k1, d1, k2, d2, img2_w, img2_h, min_level=0, max_level=1, radius=10.0, th_low=50,
                           nratio=0.0, histo_len=30, check_orientation=True):
        k1 = np.ascontiguousarray(k1, KEYPOINT); k2 = np.ascontiguousarray(k2, KEYPOINT)
        d1, d2 = self._desc(d1), self._desc(d2)
        out = np.zeros(max(len(k1), 1), MATCH)
        n = C.c_int(0)
        self.check(lib().tb_search_by_violence(self._h, _p(k1), _p(d1), len(k1), _p(k2), _p(d2), len(k2), int(img2_w),
                                               int(img2_h), int(min_level), int(max_level), C.c_float(radius), int(th_low),
                                               C.c_float(nratio), int(histo_len), int(check_orientation), _p(out), len(out),
                                               C.byref(n)))
        return out[:n.value].copy()

    @staticmethod
    def _fv(fv):
        nodes = np.array(sorted(fv), np.uint32)
        start = np.zeros(len(nodes) + 1, np.int32)
        items = []
        for i, nd in enumerate(nodes):
            items.extend(int(x) for x in fv[int(nd)])
            start[i + 1] = len(items)
        return nodes, start, np.array(items, np.uint32)

    def search_by_bow(self, k1, d1, fv1, k2, d2, fv2, has_mp2=None, map_point_only=False, th_low=50, nratio=0.0, histo_len=30,
                      check_orientation=True):
        """Matcher::searchByBow(F1, F2, MapPointOnly) (reference matcher.cpp:619-721); fv1 / fv2: the frames' DBoW2 feature
        vectors as dicts {node id: [feature indices]}."""
        k1 = np.ascontiguousarray(k1, KEYPOINT); k2 = np.ascontiguousarray(k2, KEYPOINT)
        d1, d2 = self._desc(d1), self._desc(d2)
        n1a, s1a, i1a = self._fv(fv1)
        n2a, s2a, i2a = self._fv(fv2)
        hm = None if has_mp2 is None else np.ascontiguousarray(has_mp2, np.uint8)
        out = np.zeros(max(len(i1a), 1), MATCH)
        n = C.c_int(0)
        self.check(lib().tb_search_by_bow(self._h, _p(k1), _p(d1), len(k1), _p(n1a), _p(s1a), _p(i1a), len(n1a), _p(k2), _p(d2), len(k2),
                                          _p(hm), _p(n2a), _p(s2a), _p(i2a), len(n2a), int(map_point_only), int(th_low),
                                          C.c_float(nratio), int(histo_len), int(check_orientation), _p(out), len(out), C.byref(n)))
        return out[:n.value].copy()

    def clahe(self, img, clip_limit=3.0, tiles=(8, 8)):
        """Frame::Equalize (reference Frame.cpp:453-458): cv::createCLAHE(3.0, Size(8, 8))->apply."""
        img = np.ascontiguousarray(img, np.uint8)
        assert img.ndim == 2
        h, w = img.shape
        out = np.zeros_like(img)
        self.check(lib().tb_clahe(self._h, _p(img), w, h, w, C.c_double(clip_limit), int(tiles[0]), int(tiles[1]), _p(out), w))
        return out

    def optical_flow_pyr_lk(self, prev, nxt, prev_pts, win=21, max_level=3):
        """cv::calcOpticalFlowPyrLK as Matcher::searchByOPFlow calls it (reference matcher.cpp:744).
        Returns (next_pts [n,2], status [n] u8, err [n], coarsest level used)."""
        prev = np.ascontiguousarray(prev, np.uint8); nxt = np.ascontiguousarray(nxt, np.uint8)
        assert prev.ndim == 2 and prev.shape == nxt.shape
        h, w = prev.shape
        pts = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2)
        n = len(pts)
        out = np.zeros((max(n, 1), 2), np.float32)
        status = np.zeros(max(n, 1), np.uint8)
        err = np.zeros(max(n, 1), np.float32)
        top = C.c_int(0)
        self.check(lib().tb_optical_flow_pyr_lk(self._h, _p(prev), _p(nxt), w, h, w, _p(pts), n, int(win), int(max_level),
                                                _p(out), _p(status), _p(err), C.byref(top)))
        return out[:n], status[:n], err[:n], top.value

    def optical_flow_pyr_lk_dev(self, prev_ptr, next_ptr, width, height, stride, pts_ptr, n, out_ptr, status_ptr, err_ptr=0,
                                win=21, max_level=3):
        """Device-resident form (all arguments are device addresses); asynchronous on the context's stream."""
        self.check(lib().tb_optical_flow_pyr_lk_dev(self._h, C.c_void_p(prev_ptr), C.c_void_p(next_ptr), int(width), int(height),
                                                    int(stride), C.c_void_p(pts_ptr), int(n), int(win), int(max_level),
                                                    C.c_void_p(out_ptr), C.c_void_p(status_ptr), C.c_void_p(err_ptr or None)))

    def optical_flow_pyr_lk_batch_dev(self, npairs, prev_ptr, next_ptr, width, height, stride, image_pitch, pts_ptr, counts_ptr,
                                      pts_pitch, out_ptr, status_ptr, err_ptr=0, win=21, max_level=3):
        """Batched device-resident form: one launch per stage for all pairs; asynchronous on the context's stream."""
        self.check(lib().tb_optical_flow_pyr_lk_batch_dev(
            self._h, int(npairs), C.c_void_p(prev_ptr), C.c_void_p(next_ptr), int(width), int(height), int(stride),
            C.c_size_t(image_pitch), C.c_void_p(pts_ptr), C.c_void_p(counts_ptr or None), int(pts_pitch), int(win), int(max_level),
            C.c_void_p(out_ptr), C.c_void_p(status_ptr), C.c_void_p(err_ptr or None)))

    def search_by_opflow_batch_dev(self, npairs, img1_ptr, img2_ptr, width, height, stride, image_pitch, cam1, keys_ptr, counts_ptr,
                                   pts_pitch, cur_ptr, status_ptr, out_ptr, cap, out_counts_ptr, equalized=False, reject=False):
        """Batched device-resident Matcher::searchByOPFlow (cam1: host CAMERA record); asynchronous on the context's stream."""
        cam1 = np.ascontiguousarray(cam1, CAMERA)
        self.check(lib().tb_search_by_opflow_batch_dev(
            self._h, int(npairs), C.c_void_p(img1_ptr), C.c_void_p(img2_ptr), int(width), int(height), int(stride),
            C.c_size_t(image_pitch), _p(cam1), C.c_void_p(keys_ptr), C.c_void_p(counts_ptr or None), int(pts_pitch), int(equalized),
            int(reject), C.c_void_p(cur_ptr), C.c_void_p(status_ptr), C.c_void_p(out_ptr), int(cap), C.c_void_p(out_counts_ptr)))

    def search_by_opflow(self, img1, img2, cam1, keys2_xy, equalized=False, reject=False):
        """Matcher::searchByOPFlow(F1, F2, cur_points, equalized, reject) (reference matcher.cpp:724-768).
        Returns (cur_points [n,2], DMatch records)."""
        img1 = np.ascontiguousarray(img1, np.uint8); img2 = np.ascontiguousarray(img2, np.uint8)
        assert img1.ndim == 2 and img1.shape == img2.shape
        h, w = img1.shape
        cam1 = np.ascontiguousarray(cam1, CAMERA)
        pts = np.ascontiguousarray(keys2_xy, np.float32).reshape(-1, 2)
        n = len(pts)
        cur = np.zeros((max(n, 1), 2), np.float32)
        out = np.zeros(max(n, 1), MATCH)
        cnt = C.c_int(0)
        self.check(lib().tb_search_by_opflow(self._h, _p(img1), _p(img2), w, h, w, _p(cam1), _p(pts), n, int(equalized),
                                             int(reject), _p(cur), _p(out), len(out), C.byref(cnt)))
        return cur[:n], out[:cnt.value].copy()

    def find_fundamental_ransac(self, pts1, pts2, thresh=1.0, conf=0.99):
        """cv::findFundamentalMat(pts1, pts2, FM_RANSAC, thresh, conf, mask) (reference matcher.cpp:872).
        Returns (ok, mask [n] u8, F [3,3] f64, RANSAC iterations run)."""
        p1 = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2)
        p2 = np.ascontiguousarray(pts2, np.float32).reshape(-1, 2)
        assert len(p1) == len(p2)
        n = len(p1)
        mask = np.zeros(max(n, 1), np.uint8)
        F = np.zeros(9, np.float64)
        it, ok = C.c_int(0), C.c_int(0)
        self.check(lib().tb_find_fundamental_ransac(self._h, _p(p1), _p(p2), n, C.c_double(thresh), C.c_double(conf), _p(mask), _p(F),
                                                    C.byref(it), C.byref(ok)))
        return ok.value, mask[:n], F.reshape(3, 3), it.value

    def reject_with_f(self, cur_pts, last_pts, status):
        """Matcher::rejectWithF(cur_pts, last_pts, status) (reference matcher.cpp:853-881). Returns the updated flags."""
        cur = np.ascontiguousarray(cur_pts, np.float32).reshape(-1, 2)
        last = np.ascontiguousarray(last_pts, np.float32).reshape(-1, 2)
        st = np.ascontiguousarray(status, np.uint8).copy()
        assert len(cur) == len(last) == len(st)
        self.check(lib().tb_reject_with_f(self._h, _p(cur), _p(last), len(st), _p(st)))
        return st

    def vocab_create(self, voc):
        """Upload a synth.Vocabulary (tb_vocabulary arrays); returns an opaque handle for bow_transform / vocab_destroy."""
        h = C.c_void_p()
        self.check(lib().tb_vocab_create(self._h, C.byref(voc.c), C.byref(h)))
        return h

    def vocab_destroy(self, h):
        lib().tb_vocab_destroy(h)

    def bow_transform(self, vocab_handle, desc, levelsup=4):
        """Frame::SetBow's voc->transform per feature (reference Frame.cpp:267-270): (word_ids, weights, node_ids)."""
        d = self._desc(desc)
        n = len(d)
        wid = np.zeros(max(n, 1), np.int32); wt = np.zeros(max(n, 1), np.float64); nid = np.zeros(max(n, 1), np.int32)
        self.check(lib().tb_bow_transform(self._h, vocab_handle, _p(d), n, int(levelsup), _p(wid), _p(wt), _p(nid)))
        return wid[:n], wt[:n], nid[:n]

    def reject_with_f_batch(self, cur, last, status, counts=None):
        """tb_reject_with_f_batch_dev on torch tensors of this context's device: cur / last float32 [P, N, 2], status uint8
        [P, N] (updated in place and returned), counts int32 [P] or None. Asynchronous on the context's stream."""
        P, N = status.shape
        assert cur.shape == (P, N, 2) and last.shape == (P, N, 2) and cur.is_contiguous() and last.is_contiguous() and status.is_contiguous()
        self.check(lib().tb_reject_with_f_batch_dev(self._h, P, C.c_void_p(cur.data_ptr()), C.c_void_p(last.data_ptr()),
                                                    C.c_void_p(counts.data_ptr()) if counts is not None else None, N,
                                                    C.c_void_p(status.data_ptr())))
        self.synchronize()
        return status

    def add_map_points_by_stereo(self, img_stereo, img_current, cam_stereo, keys_xy, bf):
        """LocalBA::AddMapPointsByStereo(current_frame, stereo_frame, bf, fx) (reference LocalBA.cpp:46-68): depth per key."""
        a = np.ascontiguousarray(img_stereo, np.uint8); b = np.ascontiguousarray(img_current, np.uint8)
        assert a.ndim == 2 and a.shape == b.shape
        h, w = a.shape
        cam = np.ascontiguousarray(cam_stereo, CAMERA)
        pts = np.ascontiguousarray(keys_xy, np.float32).reshape(-1, 2)
        n = len(pts)
        depth = np.zeros(max(n, 1), np.float32)
        cnt = C.c_int(0)
        self.check(lib().tb_add_map_points_by_stereo(self._h, _p(a), _p(b), w, h, w, _p(cam), _p(pts), n, C.c_float(bf), _p(depth),
                                                     C.byref(cnt)))
        return depth[:n], cnt.value

    def search_by_projection(self, Tcw1, cam1, img1_w, img1_h, k1, d1, taken1, k2, mp2, mp2_desc, scale_factors, nratio,
                             th_high=100, histo_len=30, check_orientation=True):
        """Matcher::searchByProjection(F1, F2) (reference matcher.cpp:406-531); mp2 / mp2_desc aligned with F2's keys."""
        Tcw1 = np.ascontiguousarray(Tcw1, np.float32).reshape(16)
        cam1 = np.ascontiguousarray(cam1, CAMERA)
        k1 = np.ascontiguousarray(k1, KEYPOINT); k2 = np.ascontiguousarray(k2, KEYPOINT)
        d1, mp2_desc = self._desc(d1), self._desc(mp2_desc)
        taken1 = np.ascontiguousarray(taken1, np.uint8)
        mp2 = np.ascontiguousarray(mp2, MAPPOINT)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        out = np.zeros(max(len(k2), 1), MATCH)
        n = C.c_int(0)
        self.check(lib().tb_search_by_projection(self._h, _p(Tcw1), _p(cam1), int(img1_w), int(img1_h), _p(k1), _p(d1), _p(taken1),
                                                 len(k1), _p(k2), _p(mp2), _p(mp2_desc), len(k2), _p(sf), len(sf),
                                                 C.c_float(nratio), int(th_high), int(histo_len), int(check_orientation),
                                                 _p(out), len(out), C.byref(n)))
        return out[:n.value].copy()

    def search_by_projection_map(self, Tcw1, cam1, img1_w, img1_h, k1, d1, taken1, mps, mp_desc, scale_factors, nratio,
                                 radio, th_high=100):
        """Matcher::searchByProjection(map, F1, radio) (reference matcher.cpp:539-617)."""
        Tcw1 = np.ascontiguousarray(Tcw1, np.float32).reshape(16)
        cam1 = np.ascontiguousarray(cam1, CAMERA)
        k1 = np.ascontiguousarray(k1, KEYPOINT)
        d1, mp_desc = self._desc(d1), self._desc(mp_desc)
        taken1 = np.ascontiguousarray(taken1, np.uint8)
        mps = np.ascontiguousarray(mps, MAPPOINT)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        out = np.zeros(max(len(mps), 1), MATCH)
        n = C.c_int(0)
        self.check(lib().tb_search_by_projection_map(self._h, _p(Tcw1), _p(cam1), int(img1_w), int(img1_h), _p(k1), _p(d1),
                                                     _p(taken1), len(k1), _p(mps), _p(mp_desc), len(mps), _p(sf), len(sf),
                                                     C.c_float(nratio), C.c_float(radio), int(th_high), _p(out), len(out),
                                                     C.byref(n)))
        return out[:n.value].copy()

    def pose_opt(self, K, Tcw, obs, outlier=None):
        K = np.ascontiguousarray(K, np.float64)
        Tcw = np.ascontiguousarray(Tcw, np.float32).reshape(16)
        obs = np.ascontiguousarray(obs, OBS)
        outl = np.zeros(len(obs), np.uint8) if outlier is None else np.ascontiguousarray(outlier, np.uint8).copy()
        out = np.zeros(16, np.float32)
        stats = np.zeros(8, np.float64)
        n = C.c_int(0)
        self.check(lib().tb_pose_opt(self._h, _p(K), _p(Tcw), _p(obs), len(obs), _p(outl), _p(out), C.byref(n), _p(stats)))
        return n.value, out.reshape(4, 4), outl, stats

    def local_ba(self, K, poses, nfixed, pts, obs, iters=10):
        K = np.ascontiguousarray(K, np.float64)
        poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 16).copy()
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3).copy()
        obs = np.ascontiguousarray(obs, BA_OBS)
        stats = np.zeros(8, np.float64)
        self.check(lib().tb_local_ba(self._h, _p(K), len(poses), int(nfixed), _p(poses), len(pts), _p(pts), _p(obs), len(obs),
                                     int(iters), _p(stats)))
        return int(stats[0]), poses.reshape(-1, 4, 4), pts, stats


class Extractor:
    """tb_extractor: batched, device-resident pyramid + ORB / FAST-grid extraction plan."""

    def __init__(self, ctx, width, height, nlevels, scale, max_images, max_target):
        self.ctx = ctx
        self.sf, self.inv_sf, self.sigma2, self.inv_sigma2 = scale_factors(nlevels, scale)
        self.width, self.height, self.nlevels = int(width), int(height), int(nlevels)
        self.max_images = int(max_images)
        self.ws, self.hs = pyramid_sizes(width, height, self.sf)
        self._h = C.c_void_p()
        ctx.check(lib().tb_extractor_create(ctx._h, self.width, self.height, self.nlevels, _p(self.sf), None, None,
                                            self.max_images, int(max_target), C.byref(self._h)))
        self._keep = None

    def close(self):
        # tb_destroy() releases every plan of its context; only destroy while the context is alive
        if self._h and self.ctx._h:
            lib().tb_extractor_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_images_host(self, images):
        images = np.ascontiguousarray(images, np.uint8)
        if images.ndim == 2:
            images = images[None]
        assert images.shape[1:] == (self.height, self.width)
        self._keep = images
        self.ctx.check(lib().tb_extractor_set_images_host(self._h, _p(images), images.shape[0], images.strides[1],
                                                          C.c_size_t(images.strides[0])))
        return images.shape[0]

    def set_images_dev(self, dev_ptr, n, stride, pitch):
        self.ctx.check(lib().tb_extractor_set_images_dev(self._h, C.c_void_p(dev_ptr), int(n), int(stride), C.c_size_t(pitch)))

    def build_pyramid(self, n):
        self.ctx.check(lib().tb_extractor_build_pyramid(self._h, int(n)))

    def get_level(self, index, level):
        out = np.zeros((int(self.hs[level]), int(self.ws[level])), np.uint8)
        self.ctx.check(lib().tb_extractor_get_level_host(self._h, int(index), int(level), _p(out), out.strides[0]))
        return out

    def orb(self, n, target, init_th, min_th, quota_mode=0, exit_keys=None):
        ek = None if exit_keys is None else np.ascontiguousarray(exit_keys, KEYPOINT)
        self.ctx.check(lib().tb_extractor_orb(self._h, int(n), int(target), C.c_float(init_th), C.c_float(min_th),
                                              int(quota_mode), _p(ek), 0 if ek is None else len(ek)))

    def fastgrid(self, n, target, threshold, occupancy=None):
        occ = None if occupancy is None else np.ascontiguousarray(occupancy, np.uint8)
        self.ctx.check(lib().tb_extractor_fastgrid(self._h, int(n), _p(self.inv_sf), int(target), C.c_float(threshold),
                                                   _p(occ), 0 if occ is None else len(occ)))

    def counts(self, n):
        c = np.zeros(int(n), np.int32)
        self.ctx.check(lib().tb_extractor_counts_host(self._h, int(n), _p(c)))
        return c

    def results(self, index, cap=None, with_desc=True):
        cap = int(cap or 65536)
        kps = np.zeros(cap, KEYPOINT)
        desc = np.zeros((cap, 32), np.uint8) if with_desc else None
        n = C.c_int(0)
        self.ctx.check(lib().tb_extractor_results_host(self._h, int(index), _p(kps), _p(desc), cap, C.byref(n)))
        return kps[:n.value].copy(), (desc[:n.value].copy() if with_desc else None)

    def results_dev(self):
        kps, desc, counts, cap = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int(0)
        self.ctx.check(lib().tb_extractor_results_dev(self._h, C.byref(kps), C.byref(desc), C.byref(counts), C.byref(cap)))
        return kps.value, desc.value, counts.value, cap.value

    def copy_results_dev(self, n, kps_ptr, desc_ptr, counts_ptr, cap):
        self.ctx.check(lib().tb_extractor_copy_results_dev(self._h, int(n), C.c_void_p(kps_ptr), C.c_void_p(desc_ptr),
                                                           C.c_void_p(counts_ptr), int(cap)))

    def candidates(self, index, level):
        cap = int(self.ws[level]) * int(self.hs[level]) // 4 + 4096
        out = np.zeros(cap, CORNER)
        n = C.c_int(0)
        self.ctx.check(lib().tb_extractor_candidates_host(self._h, int(index), int(level), _p(out), cap, C.byref(n)))
        return out[:n.value].copy()
