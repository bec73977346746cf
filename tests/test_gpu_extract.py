"""GPU parity tests (pytest -m gpu): pyramid / FAST / quadtree / ORB / FAST-grid extraction through the
C ABI against the CPU oracle and the committed golden vectors.  Bit-exact bar (integer / byte / index
work; keypoint floats are exact products of small integers and float32 scale factors)."""
import numpy as np
import pytest

import oracle
from trackingbench_slam_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def _eq_struct(a, b):
    assert a.dtype == b.dtype and a.shape == b.shape, (a.shape, b.shape)
    for f in a.dtype.names:
        assert np.array_equal(a[f], b[f]), f


@pytest.mark.parametrize("shape,nl,scale", [((376, 1241), 5, 0.8), ((480, 640), 8, 0.8), ((300, 400), 4, 0.5),
                                             ((241, 333), 5, 0.6)])
def test_pyramid_bit_exact(ctx, kitti_pair, shape, nl, scale):
    img = kitti_pair[0] if shape == (376, 1241) else synth.frame(21, shape[1], shape[0])
    got, sf = ctx.pyramid(img, nl, scale)
    exp, sfo = oracle.pyramid(img, nl, scale)
    assert np.array_equal(sf, sfo)
    for l, (g, e) in enumerate(zip(got, exp)):
        assert g.shape == e.shape
        assert np.array_equal(g, e), "level %d" % l


@pytest.mark.parametrize("th,nms", [(20, True), (7, True), (80, True), (30, False), (0, True)])
def test_fast_detect_bit_exact(ctx, kitti_pair, th, nms):
    for img in (kitti_pair[0][:200, :300].copy(), synth.frame(22, 320, 240), synth.frame(23, 61, 64)):
        _eq_struct(ctx.fast_detect(img, th, nms), oracle.fast9(img, th, nms))


def test_fast_detect_degenerate(ctx):
    for shape in ((6, 50), (50, 6), (7, 7), (8, 9)):
        img = np.zeros(shape, np.uint8)
        img[shape[0] // 2, shape[1] // 2] = 255
        _eq_struct(ctx.fast_detect(img, 10), oracle.fast9(img, 10))
    flat = np.full((100, 100), 77, np.uint8)
    assert len(ctx.fast_detect(flat, 1)) == 0


def test_cell_candidates_bit_exact(ctx, kitti_pair, golden):
    L = kitti_pair[0]
    ex = capi.Extractor(ctx, 1241, 376, 5, 0.8, 1, 1000)
    ex.set_images_host(L)
    ex.build_pyramid(1)
    ex.orb(1, 1000, 80, 30)
    for l in range(5):
        _eq_struct(ex.candidates(0, l), golden["c5_cand_left_L%d" % l])
    ex.close()


def test_orb_extract_kitti_golden(ctx, kitti_pair, golden):
    for img, side in zip(kitti_pair, ("left", "right")):
        for tag, nl, N in (("c5", 5, 1000), ("c8", 8, 2000)):
            lv, sf = ctx.pyramid(img, nl, 0.8)
            k, d, q = ctx.orb_extract(lv, sf, N, 80, 30)
            assert np.array_equal(q, golden[f"{tag}_quotas"])
            _eq_struct(k, golden[f"{tag}_kps_{side}"])
            assert np.array_equal(d, golden[f"{tag}_desc_{side}"])


def test_orb_addpoints_golden(ctx, kitti_pair, golden):
    lv, sf = ctx.pyramid(kitti_pair[0], 5, 0.8)
    k, d, q = ctx.orb_extract(lv, sf, 1000, 80, 30)
    ka, da, _ = ctx.orb_extract(lv, sf, 1000, 80, 30, exit_keys=k, quotas=q)
    _eq_struct(ka, golden["c5_addpoints_kps_left"])
    assert np.array_equal(da, golden["c5_addpoints_desc_left"])


@pytest.mark.parametrize("seed,w,h,nl,N,ith,mth", [(31, 640, 480, 8, 1000, 80, 30), (32, 640, 480, 8, 1000, 20, 7),
                                                    (33, 320, 200, 4, 300, 40, 10), (34, 1280, 720, 8, 2000, 80, 30),
                                                    (35, 200, 320, 3, 150, 30, 10), (36, 97, 131, 2, 40, 20, 5)])
def test_orb_extract_synthetic_vs_oracle(ctx, seed, w, h, nl, N, ith, mth):
    img = synth.frame(seed, w, h)
    lv, sf = oracle.pyramid(img, nl, 0.8)
    ko, do, qo = oracle.orb_extract(lv, sf, N, ith, mth)
    k, d, q = ctx.orb_extract(lv, sf, N, ith, mth)
    assert np.array_equal(q, qo)
    _eq_struct(k, ko)
    assert np.array_equal(d, do)
    assert len(k) > 0


def test_orb_extract_batched_plan(ctx):
    imgs = np.stack([synth.frame(40 + i, 640, 480) for i in range(3)])
    ex = capi.Extractor(ctx, 640, 480, 8, 0.8, 4, 1000)
    n = ex.set_images_host(imgs)
    ex.build_pyramid(n)
    ex.orb(n, 1000, 80, 30)
    cnt = ex.counts(n)
    for b in range(n):
        lv, sf = oracle.pyramid(imgs[b], 8, 0.8)
        for l in range(8):
            assert np.array_equal(ex.get_level(b, l), lv[l])
        ko, do, _ = oracle.orb_extract(lv, sf, 1000, 80, 30)
        k, d = ex.results(b)
        assert cnt[b] == len(ko)
        _eq_struct(k, ko)
        assert np.array_equal(d, do)
    ex.close()


def test_orb_extract_batches_take_the_image_per_xcd_order(ctx):
    """From 64 images per launch on, the pyramid, FAST and descriptor kernels map workgroups to (image, tile) so that XCD k
    works through images k, k + 8, ... (what bench.py's 1024-image launches run). 70 images -- not a multiple of 8: the last
    group's missing images are workgroups that exit -- of 5 distinct frames, every slot against the oracle: pyramid levels,
    FAST candidates per level, keypoints and descriptors."""
    w, h, nl, N = 320, 200, 5, 400
    distinct = [synth.frame(90 + i, w, h) for i in range(5)]
    n = 70
    imgs = np.stack([distinct[i % 5] for i in range(n)])
    ex = capi.Extractor(ctx, w, h, nl, 0.8, n, N)
    assert ex.set_images_host(imgs) == n
    ex.build_pyramid(n)
    ex.orb(n, N, 40, 12)
    cnt = ex.counts(n)
    ref = []
    for img in distinct:
        lv, sf = oracle.pyramid(img, nl, 0.8)
        ko, do, _ = oracle.orb_extract(lv, sf, N, 40, 12)
        ref.append((lv, ko, do, [oracle.orb_candidates(lv[l], 40, 12) for l in range(nl)]))
    for b in range(n):
        lv, ko, do, cands = ref[b % 5]
        k, d = ex.results(b)
        assert cnt[b] == len(ko)
        _eq_struct(k, ko)
        assert np.array_equal(d, do)
        if b in (0, 7, 8, 63, 64, 69):             # levels and candidates: first / last of an XCD group, last image
            for l in range(nl):
                assert np.array_equal(ex.get_level(b, l), lv[l])
                _eq_struct(ex.candidates(b, l), cands[l])
    ex.close()


def test_orb_extract_edge_cases(ctx):
    sf = oracle.scale_factors(3, 0.8)[0]
    flat = [np.full((120, 160), 90, np.uint8), np.full((96, 128), 90, np.uint8), np.full((76, 102), 90, np.uint8)]
    k, d, q = ctx.orb_extract(flat, sf, 500, 80, 30)
    assert len(k) == 0
    tiny = [np.zeros((40, 40), np.uint8)] * 3  # no 30-px cell fits
    k, d, q = ctx.orb_extract(tiny, sf, 500, 80, 30)
    assert len(k) == 0
    with pytest.raises(capi.TBError) as e:  # AddPoints before operator(): TB_ESTATE
        ex = capi.Extractor(ctx, 160, 120, 3, 0.8, 1, 100)
        ex.set_images_host(flat[0])
        ex.orb(1, 100, 80, 30, quota_mode=1)
    assert e.value.code == capi.TB_ESTATE
    # quota far above the candidate count: every candidate survives alone in a leaf
    img = synth.frame(37, 160, 120)
    lv, sf2 = oracle.pyramid(img, 2, 0.8)
    ko, do, _ = oracle.orb_extract(lv, sf2, 2000, 40, 10)
    k, d, _ = ctx.orb_extract(lv, sf2, 2000, 40, 10)
    _eq_struct(k, ko)
    assert np.array_equal(d, do)
    # tiny quota: fewer leaves than initial nodes allow
    ko, do, _ = oracle.orb_extract(lv, sf2, 3, 40, 10)
    k, d, _ = ctx.orb_extract(lv, sf2, 3, 40, 10)
    _eq_struct(k, ko)
    assert np.array_equal(d, do)


def test_fastgrid_extract(ctx, kitti_pair, golden):
    lv, sf = oracle.pyramid(kitti_pair[0], 5, 0.8)
    isf = oracle.scale_factors(5, 0.8)[1]
    got = ctx.fastgrid_extract(lv, isf, 1000, 20.0)
    _eq_struct(got, golden["c5_fastgrid_left"])
    img = synth.frame(38, 640, 480)
    lv, _ = oracle.pyramid(img, 3, 0.8)
    isf = oracle.scale_factors(3, 0.8)[1]
    occ = np.zeros(2000, np.uint8); occ[::3] = 1
    for target, th, o in ((1000, 20.0, None), (200, 5.0, None), (500, 10.0, occ)):
        _eq_struct(ctx.fastgrid_extract(lv, isf, target, th, o), oracle.fastgrid_extract(lv, isf, target, th, o))


def test_config_4k_stereo_8000_keypoints(ctx):
    """BASELINE.json configs[4] geometry: 3840x2160, 8 levels, N=8000 (quadtree lists near the LDS capacity)."""
    img = synth.frame(50, 3840, 2160)
    ex = capi.Extractor(ctx, 3840, 2160, 8, 0.8, 1, 8000)
    ex.set_images_host(img)
    ex.build_pyramid(1)
    ex.orb(1, 8000, 80, 30)
    k, d = ex.results(0, 9000)
    lv, sf = oracle.pyramid(img, 8, 0.8)
    for l in (1, 4, 7):
        assert np.array_equal(ex.get_level(0, l), lv[l])
    ko, do, _ = oracle.orb_extract(lv, sf, 8000, 80, 30)
    _eq_struct(k, ko)
    assert np.array_equal(d, do)
    assert len(k) >= 8000
    ex.close()


def test_config_640x480_mono_1000_keypoints(ctx):
    """BASELINE.json configs[1]: 640x480 mono, 8 levels, 1000 keypoints, extract + match between two frames."""
    a, b = synth.frame(60, 640, 480, stereo=True)
    ra, rb = [], []
    for img in (a, b):
        lv, sf = ctx.pyramid(img, 8, 0.8)
        k, d, _ = ctx.orb_extract(lv, sf, 1000, 80, 30)
        lvo, _ = oracle.pyramid(img, 8, 0.8)
        ko, do, _ = oracle.orb_extract(lvo, sf, 1000, 80, 30)
        _eq_struct(k, ko)
        assert np.array_equal(d, do)
        ra.append(d)
    m = ctx.search_by_bf(ra[0], ra[1], 10, 30)
    _eq_struct(m, oracle.search_by_bf(ra[0], ra[1], 10, 30))


def _cand_vs_oracle(ctx, img, nl, scale, ith, mth, N=500):
    """FAST candidates of every level after the cell loop (a4), cell-major raster order, against the oracle."""
    h, w = img.shape
    ex = capi.Extractor(ctx, w, h, nl, scale, 1, N)
    ex.set_images_host(img)
    ex.build_pyramid(1)
    ex.orb(1, N, ith, mth)
    lv, _ = oracle.pyramid(img, nl, scale)
    total = 0
    for l in range(nl):
        exp = oracle.orb_candidates(lv[l], ith, mth)
        _eq_struct(ex.candidates(0, l), exp)
        total += len(exp)
    ex.close()
    return total


@pytest.mark.parametrize("seed,w,h,nl,scale,ith,mth", [(61, 1280, 720, 8, 0.8, 80, 30), (62, 640, 480, 6, 0.8, 20, 7),
                                                        (63, 333, 241, 4, 0.7, 40, 40), (64, 150, 97, 2, 0.5, 30, 50),
                                                        (65, 1241, 376, 5, 0.8, 255, 0), (66, 400, 300, 3, 0.8, 0, 0)])
def test_cell_candidates_block_kernel(ctx, seed, w, h, nl, scale, ith, mth):
    """k_fast_blocks walks blocks of up to 4 x 2 cells: levels whose cell grid is not a multiple of the block, one-cell
    levels, equal / inverted / extreme thresholds (retry at minTh only where initTh leaves nothing after the NMS)."""
    assert _cand_vs_oracle(ctx, synth.frame(seed, w, h), nl, scale, ith, mth) > 0 or ith == 255


def test_cell_candidates_dense_images(ctx):
    """Uniform noise at low thresholds: thousands of corners per block overflow the kernel's record / pixel / corner
    lists, the block is redone by the any-density path (fb_dense) -- same candidates as the oracle."""
    rng = np.random.default_rng(7)
    noise = rng.integers(0, 256, (200, 330), dtype=np.uint8)
    assert _cand_vs_oracle(ctx, noise, 3, 0.8, 12, 4) > 3000
    assert _cand_vs_oracle(ctx, noise, 2, 0.8, 3, 1) > 3000
    # half noise, half smooth: overflowing and ordinary blocks side by side
    mixed = synth.frame(67, 660, 200)
    mixed[:, :300] = rng.integers(0, 256, (200, 300), dtype=np.uint8)
    assert _cand_vs_oracle(ctx, mixed, 3, 0.8, 20, 7) > 3000


def test_cell_candidates_extreme_values(ctx):
    """Pixels of 0 and 255 only: every ring difference is 0 or +-255, the ends of the range the packed half-precision score
    routine maps onto [1281, 1791] (fb_score1); scores of 254 and single-pixel structures next to saturated neighbours."""
    rng = np.random.default_rng(11)
    blobs = (rng.random((240, 400)) < 0.5).astype(np.uint8) * 255
    blobs = np.kron(blobs[::4, ::4], np.ones((4, 4), np.uint8))          # 4 x 4 blocks of 0 / 255
    assert _cand_vs_oracle(ctx, blobs, 3, 0.8, 80, 30) > 100
    assert _cand_vs_oracle(ctx, blobs, 2, 0.8, 254, 200) >= 0            # only full-range corners survive the first pass
    specks = np.zeros((200, 330), np.uint8)
    specks[rng.integers(3, 197, 600), rng.integers(3, 327, 600)] = 255   # isolated bright pixels on black, and the inverse
    assert _cand_vs_oracle(ctx, specks, 2, 0.8, 100, 20) > 100
    assert _cand_vs_oracle(ctx, 255 - specks, 2, 0.8, 100, 20) > 100


def test_orb_extract_aligned_device_frames_with_padding(ctx):
    """Caller-owned level-0 frames with a 16-byte aligned stride wider than the image take the bounded buffer-load path of
    the FAST kernel on memory the extractor does not own: the padding (255) right of the image is loaded into tile columns no
    stage looks at, and the last row's loads stop at the end of the allocation."""
    import torch
    w, h, stride = 333, 241, 336
    img = synth.frame(70, w, h)
    buf = np.full((h, stride), 255, np.uint8)
    buf[:, :w] = img
    dev = torch.from_numpy(buf).cuda()
    ex = capi.Extractor(ctx, w, h, 4, 0.8, 1, 500)
    torch.cuda.synchronize()
    ex.set_images_dev(dev.data_ptr(), 1, stride, stride * h)
    ex.build_pyramid(1)
    ex.orb(1, 500, 40, 10)
    k, d = ex.results(0, 1000)
    lv, sf = oracle.pyramid(img, 4, 0.8)
    ko, do, _ = oracle.orb_extract(lv, sf, 500, 40, 10)
    _eq_struct(k, ko)
    assert np.array_equal(d, do)
    for l in range(4):
        _eq_struct(ex.candidates(0, l), oracle.orb_candidates(lv[l], 40, 10))
    ex.close()


def test_cell_candidates_forced_dense_path(ctx, kitti_pair):
    """tb_debug_force_dense_fast sends every block down the list-free path: it must reproduce the ordinary path's output."""
    ctx.force_dense_fast(True)
    try:
        _cand_vs_oracle(ctx, kitti_pair[0], 5, 0.8, 80, 30)
        _cand_vs_oracle(ctx, synth.frame(68, 640, 480), 8, 0.8, 20, 7)
    finally:
        ctx.force_dense_fast(False)
    _cand_vs_oracle(ctx, kitti_pair[0], 5, 0.8, 80, 30)


def test_orb_extract_unaligned_device_frames(ctx):
    """Caller-owned level-0 frames whose row stride is not a multiple of 16 bytes take the byte-wise tile load."""
    import torch
    w, h, stride = 333, 241, 341
    img = synth.frame(69, w, h)
    buf = np.zeros((h, stride), np.uint8)
    buf[:, :w] = img
    buf[:, w:] = 255                      # padding must never be read as pixels
    dev = torch.from_numpy(buf).cuda()
    ex = capi.Extractor(ctx, w, h, 4, 0.8, 1, 500)
    torch.cuda.synchronize()
    ex.set_images_dev(dev.data_ptr(), 1, stride, stride * h)
    ex.build_pyramid(1)
    ex.orb(1, 500, 40, 10)
    k, d = ex.results(0, 1000)
    lv, sf = oracle.pyramid(img, 4, 0.8)
    ko, do, _ = oracle.orb_extract(lv, sf, 500, 40, 10)
    _eq_struct(k, ko)
    assert np.array_equal(d, do)
    for l in range(4):
        _eq_struct(ex.candidates(0, l), oracle.orb_candidates(lv[l], 40, 10))
    ex.close()
