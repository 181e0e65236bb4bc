"""CPU checks of the post-process oracle (oracle/postprocess_ref.py): published 8-bit OpenCV
known answers and structural properties.  (Parity of this stage is UNPINNED: no cv2 here.)"""
import numpy as np
import pytest

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


def test_colour_conversions_track_their_float_definitions():
    """Independent of OpenCV's fixed-point tables: the 8-bit conversions of the oracle must stay within a small bound of
    the DEFINITIONS OpenCV documents for them (cvtColor docs), evaluated in float64 -- sRGB -> linear (IEC 61966-2-1) ->
    XYZ (D65) -> CIE L*a*b*, 8-bit encoding L*255/100, a+128, b+128; HSV with H in [0,180).  OpenCV's own 8-bit path is a
    fixed-point approximation of the same definitions (documented error <= 2 LSB-ish), so a restatement that drifts
    further than that is wrong whatever cv2 would say.  200k random colours + the cube's corners and the gray axis."""
    import colorsys
    rng = np.random.default_rng(11)
    cols = rng.integers(0, 256, size=(200_000, 3), dtype=np.uint8)
    corners = np.array([[r, g, b] for r in (0, 255) for g in (0, 255) for b in (0, 255)], np.uint8)
    gray = np.repeat(np.arange(256, dtype=np.uint8)[:, None], 3, axis=1)
    cols = np.concatenate([cols, corners, gray])[None]
    # --- Lab
    c = cols[0].astype(np.float64) / 255.0
    lin = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)
    M = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
    xyz = lin @ M.T
    xyz[:, 0] /= 0.950456
    xyz[:, 2] /= 1.088754
    f = np.where(xyz > 0.008856, np.cbrt(xyz), 7.787 * xyz + 16.0 / 116.0)
    L = np.where(xyz[:, 1] > 0.008856, 116.0 * np.cbrt(xyz[:, 1]) - 16.0, 903.3 * xyz[:, 1])
    a = 500.0 * (f[:, 0] - f[:, 1])
    b = 200.0 * (f[:, 1] - f[:, 2])
    want = np.stack([L * 255.0 / 100.0, a + 128.0, b + 128.0], axis=1)
    got = pp.rgb2lab_u8(cols)[0].astype(np.float64)
    d = np.abs(got - np.clip(want, 0, 255))
    # the fixed-point gamma / cube-root tables are coarse where the curves are steep: a few colours (dark ones mostly, e.g. RGB (1, 9, 57))
    # deviate by up to 2.4 in a / b; 99.5 % of colours are the rounded float value
    assert d.max() <= 3.0, (float(d.max()), cols[0][np.unravel_index(d.argmax(), d.shape)[0]].tolist())
    assert (d <= 0.75).mean() > 0.99 and (d <= 1.5).mean() > 0.9995, (float((d <= 0.75).mean()), float((d <= 1.5).mean()))
    # and back: Lab2RGB of an 8-bit Lab triple against the float inverse of the SAME triple (the round trip itself is lossy by up
    # to ~25 levels for saturated dark colours: 8-bit a / b quantisation, not an implementation matter)
    lab8 = pp.rgb2lab_u8(cols)
    l8 = lab8[0].astype(np.float64)
    Ls, as_, bs = l8[:, 0] * 100.0 / 255.0, l8[:, 1] - 128.0, l8[:, 2] - 128.0
    fy = (Ls + 16.0) / 116.0
    Y = np.where(Ls > 903.3 * 0.008856, fy ** 3, Ls / 903.3)
    fy = np.where(Y > 0.008856, np.cbrt(Y), 7.787 * Y + 16.0 / 116.0)
    fx, fz = fy + as_ / 500.0, fy - bs / 200.0
    inv = lambda t: np.where(t > 6.0 / 29.0, t ** 3, (t - 16.0 / 116.0) / 7.787)
    X, Z = inv(fx) * 0.950456, inv(fz) * 1.088754
    lin2 = np.stack([X, Y, Z], axis=1) @ np.linalg.inv(M).T
    lin2 = np.clip(lin2, 0.0, 1.0)
    srgb = np.where(lin2 <= 0.0031308, 12.92 * lin2, 1.055 * lin2 ** (1 / 2.4) - 0.055) * 255.0
    back = pp.lab2rgb_u8(lab8)[0].astype(np.float64)
    e = np.abs(back - np.clip(srgb, 0, 255))
    assert (e <= 1.0).mean() > 0.99 and e.max() <= 4.0, (float(e.max()), float((e <= 1.0).mean()))
    # --- HSV
    hsv = pp.rgb2hsv_u8(cols)[0].astype(np.float64)
    cf = cols[0].astype(np.float64)
    V = cf.max(axis=1)
    mn = cf.min(axis=1)
    S = np.where(V > 0, 255.0 * (V - mn) / np.maximum(V, 1), 0.0)
    assert np.array_equal(hsv[:, 2], V) and np.abs(hsv[:, 1] - S).max() <= 0.6
    hh = np.array([colorsys.rgb_to_hsv(*(px / 255.0))[0] * 180.0 for px in cf[:20000]])
    dh = np.abs(hsv[:20000, 0] - hh)
    dh = np.minimum(dh, 180.0 - dh)
    sat = (V[:20000] - mn[:20000]) > 0
    assert dh[sat].max() <= 0.75, float(dh[sat].max())     # 12-bit reciprocal tables: up to ~0.6 of a level from the exact hue


def test_gaussian_blur_tracks_the_float_definition():
    """The fixed-point blur against scipy's float Gaussian with the same support (ksize = cvRound(6 sigma + 1) | 1) and
    BORDER_REFLECT_101 (scipy mode='mirror')."""
    from scipy import ndimage
    rng = np.random.default_rng(12)
    img = rng.integers(0, 256, size=(61, 83, 3), dtype=np.uint8)
    for sigma in (1.2, 1.5):
        n = int(np.rint(sigma * 6 + 1)) | 1
        x = np.arange(n) - (n - 1) / 2
        k = np.exp(-x * x / (2 * sigma * sigma))
        k /= k.sum()
        ref = img.astype(np.float64)
        ref = ndimage.correlate1d(ref, k, axis=0, mode="mirror")
        ref = ndimage.correlate1d(ref, k, axis=1, mode="mirror")
        d = np.abs(pp.gaussian_blur_u8(img, sigma).astype(np.float64) - ref)
        # taps quantised to 1/256 (each up to ~0.9/256 off) on full-range noise: up to ~1.3 levels from the float blur, 0.3 on average
        assert d.max() <= 1.6 and d.mean() <= 0.45, (sigma, float(d.max()), float(d.mean()))


def test_oracle_against_cv2_golden(golden_dir):
    """The pin this oracle lacks (no cv2 in the build image, no reference-held fixture): when someone with an
    OpenCV wheel has run tools/make_cv2_golden.py, tests/golden/g9_cv2_postprocess.npz holds every stage of the reference's
    cv2 call chain (wow_sr.py:190-207, farm_sr.py:66-106) on this repo's test images and on the reference's own upload.  Each
    oracle stage is then fed cv2's own previous stage (so a difference is not carried along) and must agree within 2 LSB; the
    exact-match rate per stage is printed.  Skipped when the file is absent -- the oracle then stays "parity unpinned"."""
    f = golden_dir / "g9_cv2_postprocess.npz"
    if not f.exists():
        pytest.skip("no cv2 golden (run tools/make_cv2_golden.py where opencv-contrib-python>=4.8.0 is installed)")
    g = np.load(f)
    names = sorted({k.split(".")[0] for k in g.files if k.endswith(".img")})
    worst = {}
    for name in names:
        img = g[f"{name}.img"]
        for tag, (clip, grid, sigma, w_img, w_blur, gain) in (("wow", (2.5, 8, 1.2, 1.4, -0.4, 1.2)), ("farm", (2.5, 8, 1.5, 2.2, -1.2, 1.3))):
            ref = {k: g[f"{name}.{tag}.{k}"] for k in ("lab", "clahe_l", "contrast", "blur", "sharp", "hsv", "final")}
            lab_cl = ref["lab"].copy()
            lab_cl[..., 0] = ref["clahe_l"]
            hsv_f = ref["hsv"].astype(np.float32)
            mask = (hsv_f[..., 0] > 35) & (hsv_f[..., 0] < 85)
            hsv_f[..., 1] = np.where(mask, np.clip(hsv_f[..., 1] * np.float32(gain), 0, 255), hsv_f[..., 1])
            got = {
                "lab": pp.rgb2lab_u8(img),
                "clahe_l": pp.clahe_u8(ref["lab"][..., 0], clip, grid),
                "contrast": pp.lab2rgb_u8(lab_cl),
                "blur": pp.gaussian_blur_u8(ref["contrast"], sigma),
                "sharp": pp.add_weighted_u8(ref["contrast"], w_img, ref["blur"], w_blur),
                "hsv": pp.rgb2hsv_u8(ref["sharp"]),
                "final": pp.hsv2rgb_u8(hsv_f.astype(np.uint8)),
            }
            for k in got:
                d = np.abs(got[k].astype(np.int16) - ref[k].astype(np.int16))
                if k == "hsv":          # hue wraps at 180
                    d[..., 0] = np.minimum(d[..., 0], 180 - d[..., 0])
                worst[k] = max(worst.get(k, 0), int(d.max()))
                print(f"cv2 {g['cv2_version']} {name}/{tag}/{k}: max |d| {int(d.max())}, identical {np.mean(d == 0):.5f}")
                assert d.max() <= 2, (name, tag, k, int(d.max()))
            end = pp.enhance_for_crops(img) if tag == "wow" else pp.farm_postprocess(img)
            d = np.abs(end.astype(np.int16) - ref["final"].astype(np.int16))
            print(f"cv2 {name}/{tag}/end-to-end: max |d| {int(d.max())}, identical {np.mean(d == 0):.5f}")
    print("worst per stage:", worst)
