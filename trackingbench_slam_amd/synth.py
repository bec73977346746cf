"""Deterministic synthetic inputs for the tracking hot path.

Frames: u8 grey images with FAST corners at a controllable density -- low-frequency value
noise + ~W*H/400 random axis-aligned and rotated rectangles (contrast uniform in [40,160]) +
+-3 uniform pixel noise; stereo right image = same rectangles shifted left by a per-rectangle
disparity in [4,64] px.  Every random draw comes from a splitmix64 stream seeded with
0xC0FFEE + frame index, so the generator needs no files and is identical on every box.

Pose-optimisation / local-BA inputs follow the recipe of the reference's only self-contained
test, test_PoseOptimization (test/test_vo.cpp:305-355): random 3-D points projected through a
known pose, with our own fixed seed.
"""
import numpy as np

_MASK = (1 << 64) - 1
_GAMMA = 0x9E3779B97F4A7C15


def splitmix64(seed, n):
    """n successive splitmix64 outputs for `seed` as uint64."""
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & _MASK) + idx * np.uint64(_GAMMA)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


class Stream:
    """Sequential reader over a splitmix64 stream."""

    def __init__(self, seed):
        self.seed = seed & _MASK
        self.pos = 0

    def u64(self, n):
        idx = np.arange(self.pos + 1, self.pos + n + 1, dtype=np.uint64)
        self.pos += n
        with np.errstate(over="ignore"):
            z = np.uint64(self.seed) + idx * np.uint64(_GAMMA)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        return z

    def uniform(self, n, lo=0.0, hi=1.0):
        u = (self.u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
        return lo + (hi - lo) * u

    def randint(self, n, lo, hi):
        """integers in [lo, hi)"""
        return (lo + (self.u64(n) % np.uint64(hi - lo)).astype(np.int64)).astype(np.int64)


def _value_noise(st, w, h, cell=96):
    gw, gh = w // cell + 2, h // cell + 2
    g = st.uniform(gw * gh, 60.0, 190.0).reshape(gh, gw)
    ys = np.arange(h) / cell
    xs = np.arange(w) / cell
    y0 = ys.astype(int); x0 = xs.astype(int)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
    return a * (1 - fy) * (1 - fx) + b * (1 - fy) * fx + c * fy * (1 - fx) + d * fy * fx


def _draw(img, cx, cy, hw, hh, ang, delta):
    h, w = img.shape
    if ang == 0.0:
        x0, x1 = int(max(cx - hw, 0)), int(min(cx + hw, w))
        y0, y1 = int(max(cy - hh, 0)), int(min(cy + hh, h))
        if x1 > x0 and y1 > y0:
            img[y0:y1, x0:x1] += delta
        return
    r = int(np.ceil(np.hypot(hw, hh))) + 1
    x0, x1 = int(max(cx - r, 0)), int(min(cx + r, w))
    y0, y1 = int(max(cy - r, 0)), int(min(cy + r, h))
    if x1 <= x0 or y1 <= y0:
        return
    yy, xx = np.mgrid[y0:y1, x0:x1]
    ca, sa = np.cos(ang), np.sin(ang)
    u = (xx - cx) * ca + (yy - cy) * sa
    v = -(xx - cx) * sa + (yy - cy) * ca
    m = (np.abs(u) <= hw) & (np.abs(v) <= hh)
    img[y0:y1, x0:x1][m] += delta


def frame(index, width, height, stereo=False, density=400):
    """Synthetic frame `index` -> u8 image (or (left, right) when stereo)."""
    st = Stream(0xC0FFEE + int(index))
    base = _value_noise(st, width, height)
    n = max(width * height // density, 1)
    cx = st.uniform(n, 0, width); cy = st.uniform(n, 0, height)
    hw = st.uniform(n, 3, 22); hh = st.uniform(n, 3, 22)
    rot = st.uniform(n) < 0.5
    ang = np.where(rot, st.uniform(n, 0.15, np.pi / 2 - 0.15), 0.0)
    delta = st.uniform(n, 40, 160) * np.where(st.uniform(n) < 0.5, -1.0, 1.0)
    disp = st.uniform(n, 4, 64)
    left = base.copy()
    for i in range(n):
        _draw(left, cx[i], cy[i], hw[i], hh[i], float(ang[i]), delta[i])
    noise_l = st.randint(width * height, -3, 4).reshape(height, width)
    out_l = np.clip(np.rint(left) + noise_l, 0, 255).astype(np.uint8)
    if not stereo:
        return out_l
    right = base.copy()
    for i in range(n):
        _draw(right, cx[i] - np.rint(disp[i]), cy[i], hw[i], hh[i], float(ang[i]), delta[i])
    noise_r = st.randint(width * height, -3, 4).reshape(height, width)
    out_r = np.clip(np.rint(right) + noise_r, 0, 255).astype(np.uint8)
    return out_l, out_r


def pose_problem(seed, n, K, noise_px=0.5, outlier_frac=0.1, nlevels=8, scale=0.8):
    """test_PoseOptimization-style inputs (test/test_vo.cpp:305-355) with a fixed seed.

    Returns (Tcw_true, Tcw_init, obs) where obs is a structured array
    (u, v, X, Y, Z, inv_sigma2) and Tcw_* are 4x4 float32.
    """
    st = Stream(0xBADC0DE + int(seed))
    fx, fy, cx, cy = K
    X = np.stack([st.uniform(n, -4, 4), st.uniform(n, -2, 2), st.uniform(n, 4, 20)], 1)
    ang = st.uniform(3, -0.08, 0.08)
    t = st.uniform(3, -0.3, 0.3)
    Rx = np.array([[1, 0, 0], [0, np.cos(ang[0]), -np.sin(ang[0])], [0, np.sin(ang[0]), np.cos(ang[0])]])
    Ry = np.array([[np.cos(ang[1]), 0, np.sin(ang[1])], [0, 1, 0], [-np.sin(ang[1]), 0, np.cos(ang[1])]])
    Rz = np.array([[np.cos(ang[2]), -np.sin(ang[2]), 0], [np.sin(ang[2]), np.cos(ang[2]), 0], [0, 0, 1]])
    R = Rz @ Ry @ Rx
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    Xf = X.astype(np.float32)
    pc = Xf.astype(np.float64) @ R.T + t
    u = pc[:, 0] / pc[:, 2] * fx + cx + st.uniform(n, -noise_px, noise_px)
    v = pc[:, 1] / pc[:, 2] * fy + cy + st.uniform(n, -noise_px, noise_px)
    bad = st.uniform(n) < outlier_frac
    u = np.where(bad, u + st.uniform(n, -60, 60), u)
    v = np.where(bad, v + st.uniform(n, -60, 60), v)
    octave = st.randint(n, 0, nlevels)
    sf = np.float32(1.0)
    sfs = [sf]
    for _ in range(1, nlevels):
        sf = np.float32(sf * np.float32(scale)); sfs.append(sf)
    sfs = np.array(sfs, np.float32)
    inv_sigma2 = (np.float32(1.0) / (sfs * sfs)).astype(np.float32)[octave]
    obs = np.zeros(n, dtype=[("u", "<f4"), ("v", "<f4"), ("X", "<f4"), ("Y", "<f4"), ("Z", "<f4"),
                             ("inv_sigma2", "<f4")])
    obs["u"], obs["v"] = u, v
    obs["X"], obs["Y"], obs["Z"] = Xf[:, 0], Xf[:, 1], Xf[:, 2]
    obs["inv_sigma2"] = inv_sigma2
    return T.astype(np.float32), np.eye(4, dtype=np.float32), obs


def ba_problem(seed, nkf, npt, K, obs_per_pt=5, noise_px=0.5, pose_noise=0.02, pt_noise=0.05):
    """Local-BA window (extension; no reference counterpart): nkf keyframes on a short arc,
    npt points, each seen by up to obs_per_pt keyframes. Returns (poses_true, poses_init, pts_true,
    pts_init, obs) with obs a structured array (kf, pt, u, v, inv_sigma2)."""
    st = Stream(0xBA0000 + int(seed))
    fx, fy, cx, cy = K
    X = np.stack([st.uniform(npt, -6, 6), st.uniform(npt, -2.5, 2.5), st.uniform(npt, 5, 25)], 1)
    poses = np.zeros((nkf, 4, 4)); poses[:] = np.eye(4)
    for k in range(nkf):
        a = 0.01 * k
        poses[k, :3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        poses[k, :3, 3] = [-0.25 * k, 0.0, -0.05 * k]
    kfs = st.randint(npt * obs_per_pt, 0, nkf).reshape(npt, obs_per_pt)
    rows = []
    noise = st.uniform(npt * obs_per_pt * 2, -noise_px, noise_px).reshape(npt, obs_per_pt, 2)
    for p in range(npt):
        for k in sorted(set(kfs[p].tolist())):
            pc = poses[k, :3, :3] @ X[p] + poses[k, :3, 3]
            if pc[2] < 0.5:
                continue
            j = int(np.where(kfs[p] == k)[0][0])
            rows.append((k, p, pc[0] / pc[2] * fx + cx + noise[p, j, 0], pc[1] / pc[2] * fy + cy + noise[p, j, 1], 1.0))
    obs = np.array(rows, dtype=[("kf", "<i4"), ("pt", "<i4"), ("u", "<f4"), ("v", "<f4"), ("inv_sigma2", "<f4")])
    order = np.lexsort((obs["kf"], obs["pt"]))
    obs = obs[order]
    poses_init = poses.copy()
    dn = st.uniform(nkf * 6, -1, 1).reshape(nkf, 6)
    for k in range(2, nkf):
        poses_init[k, :3, 3] += pose_noise * dn[k, :3]
        a = pose_noise * 0.2 * dn[k, 3]
        Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        poses_init[k, :3, :3] = Rz @ poses_init[k, :3, :3]
    pts_init = X + pt_noise * st.uniform(npt * 3, -1, 1).reshape(npt, 3)
    return (poses.astype(np.float32), poses_init.astype(np.float32), X.astype(np.float32),
            pts_init.astype(np.float32), obs)


def projection_case(seed, n1=1500, nmp=1200, width=1241, height=376, nlevels=8, scale=0.8, K=(718.856, 718.856, 607.1928, 185.2157),
                    distortion=None):
    """Inputs of the projection matchers: a current frame F1 (pose, keys, descriptors, taken
    flags) and nmp map points with reference-frame keys F2 aligned to them. About 70 % of the map points project
    into F1 next to one of its keys (descriptor = that key's with a few flipped bits, octave within one level, angle
    rotated by a common offset), the rest are behind the camera, outside the image, bad, or unmatched.
    Returns a dict of arrays in the C-ABI layouts (capi.KEYPOINT / MAPPOINT / CAMERA)."""
    from . import capi
    st = Stream(0x9207EC7 + int(seed))
    fx, fy, cx, cy = K
    cam = np.zeros(1, capi.CAMERA)
    cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["width"], cam["height"] = fx, fy, cx, cy, width, height
    if distortion is not None:
        cam["has_distortion"] = 1
        cam["d"][0] = np.asarray(distortion, np.float32)
    a = 0.03
    T = np.eye(4)
    T[:3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    T[:3, 3] = [0.3, -0.05, 0.2]
    T = T.astype(np.float32)
    sf = np.ones(nlevels, np.float32)
    for i in range(1, nlevels):
        sf[i] = sf[i - 1] * np.float32(scale)
    # F1 keys
    k1 = np.zeros(n1, capi.KEYPOINT)
    k1["x"] = st.uniform(n1, 4, width - 4).astype(np.float32)
    k1["y"] = st.uniform(n1, 4, height - 4).astype(np.float32)
    k1["octave"] = st.randint(n1, 0, nlevels)
    k1["angle"] = st.uniform(n1, 0, 360).astype(np.float32)
    k1["size"] = 31.0
    d1 = (st.u64(n1 * 4).view(np.uint8)).reshape(n1, 32).copy()
    taken1 = (st.uniform(n1) < 0.1).astype(np.uint8)
    # map points: back-project a key of F1 (or a random pixel) at a random depth, world = Twc * Pc
    R, t = T[:3, :3].astype(np.float64), T[:3, 3].astype(np.float64)
    src = st.randint(nmp, 0, n1)
    kind = st.uniform(nmp)                      # < .7 near a key, < .8 random pixel, < .9 behind, else outside
    depth = st.uniform(nmp, 4.0, 40.0)
    px = np.where(kind < 0.7, k1["x"][src] + st.uniform(nmp, -2.5, 2.5), st.uniform(nmp, 0, width))
    py = np.where(kind < 0.7, k1["y"][src] + st.uniform(nmp, -2.5, 2.5), st.uniform(nmp, 0, height))
    px = np.where(kind >= 0.9, px + 3 * width, px)
    depth = np.where((kind >= 0.8) & (kind < 0.9), -depth, depth)
    Pc = np.stack([(px - cx) / fx * depth, (py - cy) / fy * depth, depth], 1)
    Pw = (Pc - t) @ R                           # R^T (Pc - t)
    Ow = -R.T @ t
    mp = np.zeros(nmp, capi.MAPPOINT)
    mp["pos"] = Pw.astype(np.float32)
    n = Ow[None, :] - Pw
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    n = -n                                      # IsInFrustum: viewCos = (P - Ow) . normal / dist
    tilt = st.uniform(nmp * 3, -0.5, 0.5).reshape(nmp, 3) * (st.uniform(nmp) < 0.3)[:, None]
    n = n + tilt
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    mp["normal"] = n.astype(np.float32)
    dist = np.linalg.norm(Pw - Ow[None, :], axis=1)
    mp["min_dist"] = (dist * np.where(st.uniform(nmp) < 0.9, 0.6, 1.2)).astype(np.float32)
    mp["max_dist"] = (dist * np.where(st.uniform(nmp) < 0.9, 1.7, 0.9)).astype(np.float32)
    mp["bad"] = (st.uniform(nmp) < 0.08).astype(np.int32)
    flips = st.randint(nmp * 12, 0, 256).reshape(nmp, 12)
    nflip = st.randint(nmp, 0, 13)
    mpd = np.where((kind < 0.7)[:, None], d1[src], (st.u64(nmp * 4).view(np.uint8)).reshape(nmp, 32)).copy()
    for i in range(nmp):
        for b in flips[i, :nflip[i]]:
            mpd[i, b >> 3] ^= np.uint8(1 << (b & 7))
    k2 = np.zeros(nmp, capi.KEYPOINT)
    k2["x"] = st.uniform(nmp, 0, width).astype(np.float32)
    k2["y"] = st.uniform(nmp, 0, height).astype(np.float32)
    k2["octave"] = np.clip(k1["octave"][src] + st.randint(nmp, -1, 2), 0, nlevels - 1)
    rot = np.where(st.uniform(nmp) < 0.8, 25.0, st.uniform(nmp, 0, 360))
    k2["angle"] = ((k1["angle"][src] + rot + st.uniform(nmp, -3, 3)) % 360.0).astype(np.float32)
    k2["size"] = 31.0
    return dict(Tcw=T, cam=cam, width=width, height=height, k1=k1, d1=d1, taken1=taken1, k2=k2, mp=mp, mp_desc=mpd, sf=sf)



# ------------------------------------------------------------------ DBoW2 vocabulary
import ctypes as _C


class TbVocabulary(_C.Structure):
    """tb_vocabulary of include/tb_types.h"""
    _fields_ = [("nnodes", _C.c_int32), ("k", _C.c_int32), ("L", _C.c_int32), ("weighting", _C.c_int32), ("scoring", _C.c_int32),
                ("child_start", _C.c_void_p), ("child_items", _C.c_void_p), ("desc", _C.c_void_p), ("word_id", _C.c_void_p),
                ("weight", _C.c_void_p)]


class Vocabulary:
    """A DBoW2 tree as flat numpy arrays + the tb_vocabulary struct that points into them (keep the object alive)."""

    def __init__(self, k, L, child_start, child_items, desc, word_id, weight, weighting=0, scoring=0):
        self.k, self.L = int(k), int(L)
        self.child_start = np.ascontiguousarray(child_start, np.int32)
        self.child_items = np.ascontiguousarray(child_items, np.int32)
        self.desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        self.word_id = np.ascontiguousarray(word_id, np.int32)
        self.weight = np.ascontiguousarray(weight, np.float64)
        self.nnodes = len(self.word_id)
        assert len(self.child_start) == self.nnodes + 1 and len(self.desc) == self.nnodes and len(self.weight) == self.nnodes
        self.c = TbVocabulary(self.nnodes, self.k, self.L, int(weighting), int(scoring), self.child_start.ctypes.data,
                              self.child_items.ctypes.data, self.desc.ctypes.data, self.word_id.ctypes.data, self.weight.ctypes.data)

    def to_text(self, path):
        """The ORBvoc-style text file TemplatedVocabulary::loadFromTextFile reads (TemplatedVocabulary.h:1338-1420): first line
        k L scoring weighting, then one line per node in id order: parent, is-leaf, the 32 descriptor bytes, the weight."""
        parent = np.zeros(self.nnodes, np.int64)
        for n in range(self.nnodes):
            parent[self.child_items[self.child_start[n]:self.child_start[n + 1]]] = n
        with open(path, "w") as f:
            f.write("%d %d %d %d\n" % (self.k, self.L, self.c.scoring, self.c.weighting))
            for n in range(1, self.nnodes):
                leaf = self.child_start[n + 1] == self.child_start[n]
                f.write("%d %d %s %r\n" % (parent[n], 1 if leaf else 0, " ".join(str(int(b)) for b in self.desc[n]), float(self.weight[n])))


def vocabulary(seed, k=10, L=4, stop_frac=0.05, ragged=0.0):
    """Seeded synthetic vocabulary: a k-ary tree of depth L built breadth-first the way DBoW2 numbers its nodes (children get the
    next free ids, so every child id exceeds its parent's), random 256-bit node descriptors that inherit most bits from the
    parent (a walk then has a meaningful nearest child), word weights in (0, 3] with a fraction `stop_frac` of stopped words
    (weight 0). `ragged` > 0 drops that fraction of the internal nodes' children and turns some nodes into early leaves, like a
    k-means tree whose clusters ran empty."""
    st = Stream(0xB0C0 + seed)
    child_start, child_items, desc, is_leaf, depth = [0], [], [np.zeros(32, np.uint8)], [False], [0]
    order = [0]
    head = 0
    while head < len(order):
        n = order[head]; head += 1
        d = depth[n]
        nchild = 0
        if d < L and not (n > 0 and ragged > 0 and st.uniform(1)[0] < ragged * 0.3):
            nchild = k if ragged <= 0 else max(1, int(k - np.floor(st.uniform(1)[0] * ragged * k)))
        ids = []
        for _ in range(nchild):
            cid = len(desc)
            flip = st.randint(256, 0, 100) < (45 if d == 0 else 12)          # bits that differ from the parent
            bits = np.unpackbits(desc[n]) ^ flip.astype(np.uint8)
            desc.append(np.packbits(bits))
            is_leaf.append(False); depth.append(d + 1)
            ids.append(cid); order.append(cid)
        # flat CSR rows must be written in node-id order: remember and assemble below
        child_items.append((n, ids))
    nn = len(desc)
    rows = {n: ids for n, ids in child_items}
    cs, ci = [0], []
    for n in range(nn):
        ci.extend(rows.get(n, []))
        cs.append(len(ci))
    word_id = np.zeros(nn, np.int32); weight = np.zeros(nn, np.float64)
    w = 0
    for n in range(nn):
        if cs[n + 1] == cs[n]:
            word_id[n] = w; w += 1
            weight[n] = 0.0 if st.uniform(1)[0] < stop_frac else 0.05 + 2.95 * float(st.uniform(1)[0])
    return Vocabulary(k, L, cs, ci, np.stack(desc), word_id, weight)


def descriptors_near_words(seed, voc, n, flips=20):
    """n descriptors: random words' descriptors with a few flipped bits"""
    st = Stream(0xD0 + seed)
    leaves = np.flatnonzero(np.diff(voc.child_start) == 0)
    pick = leaves[st.randint(n, 0, len(leaves))]
    bits = np.unpackbits(voc.desc[pick], axis=1)
    mask = st.randint(n * 256, 0, 256).reshape(n, 256) < flips
    return np.packbits(bits ^ mask.astype(np.uint8), axis=1)
