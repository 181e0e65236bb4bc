"""CPU checks of the post-process oracle (oracle/postprocess_ref.py): published 8-bit OpenCV
known answers and structural properties.  (Parity of this stage is UNPINNED: no cv2 here.)"""
import numpy as np

from oracle import postprocess_ref as pp


def test_known_answers_lab_hsv():
    cols = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [255, 255, 0]]], np.uint8)
    lab = pp.rgb2lab_u8(cols)[0].tolist()
    # cv2.cvtColor(..., COLOR_RGB2LAB) on uint8 primaries (widely published values)
    assert lab[:5] == [[136, 208, 195], [224, 42, 211], [82, 207, 20], [255, 128, 128], [0, 128, 128]]
    hsv = pp.rgb2hsv_u8(cols)[0].tolist()
    assert hsv == [[0, 255, 255], [60, 255, 255], [120, 255, 255], [0, 0, 255], [0, 0, 0], [30, 255, 255]]
    assert np.array_equal(pp.hsv2rgb_u8(pp.rgb2hsv_u8(cols)), cols)


def test_gray_axis_roundtrip_and_identity_properties():
    g = np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, axis=2)
    lab = pp.rgb2lab_u8(g)
    assert np.all(lab[..., 1] == 128) and np.all(lab[..., 2] == 128)
    assert np.all(np.diff(lab[0, :, 0].astype(int)) >= 0)            # L monotone in gray level
    assert np.abs(pp.lab2rgb_u8(lab).astype(int) - g).max() <= 1
    assert np.array_equal(pp.hsv2rgb_u8(pp.rgb2hsv_u8(g)), g)        # S == 0 path is exact


def test_gaussian_kernel_and_blur():
    for sigma, n in ((1.2, 9), (1.5, 11), (1.0, 7)):
        t = pp.gaussian_kernel_q8(sigma)
        assert len(t) == n and t.sum() == 256 and np.array_equal(t, t[::-1])
    flat = np.full((20, 30, 3), 77, np.uint8)
    assert np.array_equal(pp.gaussian_blur_u8(flat, 1.2), flat)
    assert np.array_equal(pp.unsharp(flat, 1.2, 1.4, -0.4), flat)


def test_vegetation_mask_bounds_and_trunc():
    # one pixel per hue 0..179 at S=100, V=200 -> boost applies to H in 36..84 only
    hsv = np.zeros((1, 180, 3), np.uint8)
    hsv[0, :, 0] = np.arange(180)
    hsv[0, :, 1] = 100
    hsv[0, :, 2] = 200
    rgb = pp.hsv2rgb_u8(hsv)
    back = pp.rgb2hsv_u8(rgb)
    out = pp.rgb2hsv_u8(pp.vegetation(rgb, 1.2))
    boosted = out[0, :, 1].astype(int) > back[0, :, 1].astype(int) + 5
    hh = back[0, :, 0].astype(int)
    assert not boosted[(hh <= 35) | (hh >= 85)].any()
    assert boosted[(hh > 36) & (hh < 84)].all()
    # f32(1.3)*10 is an exact tie that rounds to 13.0 -> astype(uint8) gives 13
    assert int(np.float32(10) * np.float32(1.3)) == 13
    assert int(np.clip(np.float32(250) * np.float32(1.2), 0, 255)) == 255


def test_clahe_properties():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (64, 96), dtype=np.uint8)
    out = pp.clahe_u8(img, 2.5, 8)
    assert out.shape == img.shape and out.dtype == np.uint8
    # per-pixel LUTs are monotone, so within one interpolation cell order is preserved
    blk = img[:4, :6].astype(int).ravel()
    ob = out[:4, :6].astype(int).ravel()
    i, j = np.argmin(blk), np.argmax(blk)
    assert ob[i] <= ob[j]
    # ragged size: both dimensions get padded (odd sizes must not crash)
    assert pp.clahe_u8(rng.integers(0, 256, (67, 101), dtype=np.uint8), 2.5, 8).shape == (67, 101)
    assert pp.clahe_u8(rng.integers(0, 256, (64, 101), dtype=np.uint8), 3.0, 8).shape == (64, 101)


def test_pipelines_run():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (72, 88, 3), dtype=np.uint8)
    a = pp.enhance_for_crops(img)
    b = pp.farm_postprocess(img)
    assert a.shape == img.shape == b.shape and a.dtype == np.uint8
    assert not np.array_equal(a, b)


def test_more_published_known_answers():
    """Further 8-bit OpenCV values that follow from its documented formulas without any rounding ambiguity
    (cvtColor docs: V = max, S = 255 (V - min) / V, H = 60 (G - B) / (V - min) [+120, +240] halved; addWeighted =
    saturate_cast<uchar>(cvRound(.)), cvRound = round-half-to-even)."""
    cols = np.array([[[0, 255, 255], [255, 0, 255], [128, 128, 128], [128, 0, 0], [0, 0, 128], [0, 128, 0], [255, 128, 0]]], np.uint8)
    hsv = pp.rgb2hsv_u8(cols)[0].tolist()
    assert hsv == [[90, 255, 255], [150, 255, 255], [0, 0, 128], [0, 255, 128], [120, 255, 128], [60, 255, 128], [15, 255, 255]]
    assert np.array_equal(pp.hsv2rgb_u8(np.array([[[90, 255, 255], [150, 255, 255], [0, 0, 128]]], np.uint8)), cols[:, :3])
    a = np.array([[[1, 3, 5], [255, 254, 0]]], np.uint8)
    z = np.zeros_like(a)
    assert pp.add_weighted_u8(a, 0.5, z, 0.0).tolist() == [[[0, 2, 2], [128, 127, 0]]]       # .5 ties go to the even neighbour
    assert pp.add_weighted_u8(a, 2.0, a, -0.5).tolist() == [[[2, 4, 8], [255, 255, 0]]]      # 1.5->2, 4.5->4, 7.5->8; saturation
    # the Lab L axis: L* = 116 (Y/Yn)^(1/3) - 16 scaled by 255/100; mid gray 128 -> Y = 0.2158 -> L* = 53.585 -> 136.6 -> 137
    lab = pp.rgb2lab_u8(np.array([[[128, 128, 128], [255, 255, 255], [0, 0, 0]]], np.uint8))[0].tolist()
    assert lab[1] == [255, 128, 128] and lab[2] == [0, 128, 128] and lab[0][1:] == [128, 128] and abs(lab[0][0] - 137) <= 1


def test_clahe_constant_plane_closed_form():
    """OpenCV's CLAHE on a CONSTANT plane has a closed form that needs no image: every tile's histogram is one spike of
    `area` counts at v; it is clipped to `clip = max(1, int(clipLimit * area / 256))`, the excess is spread as
    `excess // 256` to every bin plus one more to bins 0, step, 2*step, ... (step = max(256 // residual, 1)) while the
    residual lasts; lut[i] = saturate(cvRound(cdf[i] * 255 / area)).  All tiles share that LUT, so the interpolation
    returns lut[v] everywhere.  (clahe.cpp: CLAHE_CalcLut_Body; this pins the clip / redistribution / LUT-scale reading
    of the oracle independently of its code.)"""
    for (H, W, clip_limit, grid) in ((64, 64, 2.5, 8), (128, 96, 2.5, 8), (1024, 1024, 2.5, 8), (72, 72, 3.0, 8), (64, 64, 40.0, 8)):
        th, tw = H // grid, W // grid
        area = th * tw
        clip = max(int(clip_limit * area / 256), 1)
        for v in (0, 1, 77, 128, 254, 255):
            hist = np.zeros(256, np.int64)
            hist[v] = min(area, clip)
            excess = area - hist[v]
            hist += excess // 256
            residual = excess - (excess // 256) * 256
            if residual:
                step = max(256 // residual, 1)
                idx = np.arange(0, 256, step)[:residual]
                hist[idx] += 1
            lut_v = int(np.clip(np.rint(np.float32(hist[:v + 1].sum()) * (np.float32(255.0) / np.float32(area))), 0, 255))
            out = pp.clahe_u8(np.full((H, W), v, np.uint8), clip_limit, grid)
            assert np.all(out == lut_v), (H, W, clip_limit, v, int(out[0, 0]), lut_v)
    # at the wow / farm setting (2.5, 8x8) on a 1024^2 plane: tile area 16384, clip 160 -> lut[v] = round((v+1)*63.375*255/16384 + 160*255/16384)
    out = pp.clahe_u8(np.full((1024, 1024), 100, np.uint8), 2.5, 8)
    assert int(out[0, 0]) == int(np.rint((160 + 101 * 63 + len(np.arange(0, 256, 256 // 96)[:96][np.arange(0, 256, 256 // 96)[:96] <= 100])) * 255.0 / 16384))


def test_gaussian_fixed_point_taps_are_pinned():
    """The 8-bit GaussianBlur path uses 8.8 fixed-point taps that sum to exactly 256.  The float kernel follows from the
    documented getGaussianKernel formula exp(-x^2 / (2 sigma^2)) / sum; the integer taps must each lie within one unit of
    256 * w and be symmetric, non-increasing from the centre, and the literal values this oracle (and the HIP kernel) use
    are written out here so that a future cv2 golden that disagrees points at THIS line."""
    want = {1.2: [0, 4, 21, 60, 86, 60, 21, 4, 0], 1.5: [0, 2, 9, 28, 55, 68, 55, 28, 9, 2, 0], 1.0: [1, 14, 62, 102, 62, 14, 1]}
    for sigma, taps in want.items():
        t = pp.gaussian_kernel_q8(sigma)
        n = int(np.rint(sigma * 6 + 1)) | 1
        x = np.arange(n) - (n - 1) / 2
        w = np.exp(-x * x / (2 * sigma * sigma))
        w = w / w.sum() * 256.0
        assert len(t) == n and t.sum() == 256 and np.array_equal(t, t[::-1])
        assert np.all(np.abs(t - w) <= 1.0), (sigma, t.tolist(), w.round(2).tolist())
        assert np.all(np.diff(t[: n // 2 + 1]) >= 0)
        assert t.tolist() == taps, (sigma, t.tolist())


def test_add_weighted_tie_cases_are_enumerated():
    """cv2.addWeighted on 8U computes a*alpha + b*beta in float and rounds half to even (cvRound).  Where that float sum is
    EXACTLY x.5 the result depends on the rounding mode and on whether the implementation rounds the products first (SIMD
    v_fma forms) -- these are the pixels a future cv2 golden has to look at first.  For the reference's weights
    (1.4, -0.4: wow_sr.py:197; 2.2, -1.2: farm_sr.py:69) no (a, b) in [0,255]^2 is an exact tie in float32 (1.4f and 0.4f are
    not dyadic), so the rounding mode cannot matter there; with dyadic weights ties exist and must go to the even neighbour."""
    a, b = np.meshgrid(np.arange(256, dtype=np.float32), np.arange(256, dtype=np.float32), indexing="ij")
    for alpha, beta in ((1.4, -0.4), (2.2, -1.2)):
        r = a * np.float32(alpha) + b * np.float32(beta)
        ties = np.abs(r - np.floor(r) - np.float32(0.5)) == 0
        assert ties.sum() == 0, (alpha, beta, int(ties.sum()))
        # nearest misses: how close any pair comes to a tie (what a float64 / fused implementation could flip)
        near = np.abs(r - np.floor(r) - 0.5).min()
        assert near > 1e-6
        # a float64 evaluation of the same weights agrees everywhere after rounding: no pixel of this stage is rounding-mode dependent
        r64 = a.astype(np.float64) * alpha + b.astype(np.float64) * beta
        assert np.array_equal(np.clip(np.rint(r), 0, 255), np.clip(np.rint(r64), 0, 255))
    r = a * np.float32(1.5) + b * np.float32(-0.5)
    ties = (r - np.floor(r)) == 0.5
    assert ties.sum() > 1000
    got = pp.add_weighted_u8(a.astype(np.uint8), 1.5, b.astype(np.uint8), -0.5)
    exp = np.clip(np.where(ties, 2 * np.round(r / 2), np.rint(r)), 0, 255).astype(np.uint8)   # ties -> even
    assert np.array_equal(got, exp)
