"""ORACLE (test infrastructure only -- never imported by the product path).

numpy restatement of the crop-visibility post-process of the reference
(`_enhance_for_crops`, server/app/wow_sr.py:187-209; farm variants server/app/farm_sr.py:61-108).

PARITY UNPINNED.  Every arithmetic step of those functions is a call into OpenCV
(`opencv-contrib-python>=4.8.0`, server/requirements.txt:29, no upper pin, not vendored under
/root/reference, not installed in this image), and the reference holds no test vector at that
boundary.  What follows restates the published OpenCV 4.x 8-bit algorithms
(imgproc/src/color_lab.cpp `RGB2Lab_b` / `Lab2RGBinteger`, clahe.cpp, smooth.dispatch.cpp
fixed-point Gaussian, core/src/arithm.cpp addWeighted, color_hsv.simd.hpp `RGB2HSV_b` /
`HSV2RGB_b`) together with the numpy steps and constants at the reference call sites.  The
few widely published 8-bit known answers (primary colours -> Lab / HSV) are checked in
tests/test_postprocess_oracle.py; a true cv2 golden is a TODO for when a cv2 wheel exists.

All functions take / return HxWx3 uint8 RGB.
"""
from __future__ import annotations

import numpy as np

# ------------------------------------------------------------------------------------------
# tables (OpenCV initLabTabs / RGB2HSV_b tables).  Computed in float64 so that the C++ side
# (csrc/postprocess.hip, same libm) produces bit-identical tables.
# ------------------------------------------------------------------------------------------
LAB_SHIFT = 12
GAMMA_SHIFT = 3
LAB_SHIFT2 = LAB_SHIFT + GAMMA_SHIFT
LAB_CBRT_TAB_SIZE_B = 256 * 3 // 2 * (1 << GAMMA_SHIFT)
INV_GAMMA_SHIFT = 12
INV_GAMMA_TAB_SIZE = 1 << INV_GAMMA_SHIFT
LAB_BASE = 1 << 14
MIN_AB = -8145


def _round_half_even(x):
    return np.rint(x)          # cvRound == nearest, ties to even


def _tables():
    i = np.arange(256, dtype=np.float64)
    x = i / 255.0
    gamma = np.where(x <= 0.04045, x / 12.92, np.power((x + 0.055) / 1.055, 2.4))
    srgb_gamma = np.clip(_round_half_even(255.0 * (1 << GAMMA_SHIFT) * gamma), 0, 65535).astype(np.int32)

    j = np.arange(LAB_CBRT_TAB_SIZE_B, dtype=np.float64)
    xc = j / (255.0 * (1 << GAMMA_SHIFT))
    f = np.where(xc < 216.0 / 24389.0, xc * (841.0 / 108.0) + 16.0 / 116.0, np.cbrt(xc))
    lab_cbrt = np.clip(_round_half_even((1 << LAB_SHIFT2) * f), 0, 65535).astype(np.int32)

    k = np.arange(INV_GAMMA_TAB_SIZE, dtype=np.float64)
    xi = k / INV_GAMMA_TAB_SIZE
    inv = np.where(xi <= 0.0031308, xi * 12.92, 1.055 * np.power(xi, 1.0 / 2.4) - 0.055)
    srgb_inv_gamma = np.clip(_round_half_even(255.0 * inv), 0, 255).astype(np.int32)

    # L -> (y, ify), fixed point BASE = 2^14
    li = np.arange(256, dtype=np.float64)
    y_lo = _round_half_even(li * LAB_BASE * 20 * 9 / (17.0 * 29 * 29 * 29))
    ify_lo = _round_half_even(LAB_BASE * (16.0 / 116.0 + li * 5 / (3.0 * 17 * 29)))
    fy = li * 100 * LAB_BASE / (255.0 * 116) + 16.0 * LAB_BASE / 116.0
    ify_hi = _round_half_even(fy)
    y_hi = _round_half_even(fy * fy * fy / (float(LAB_BASE) * LAB_BASE))
    lab_to_y = np.where(li <= 20, y_lo, y_hi).astype(np.int64)
    lab_to_ify = np.where(li <= 20, ify_lo, ify_hi).astype(np.int64)

    sdiv = np.zeros(256, np.int64)
    hdiv = np.zeros(256, np.int64)
    n = np.arange(1, 256, dtype=np.float64)
    sdiv[1:] = _round_half_even((255 << 12) / n).astype(np.int64)
    hdiv[1:] = _round_half_even((180 << 12) / (6.0 * n)).astype(np.int64)
    return srgb_gamma, lab_cbrt, srgb_inv_gamma, lab_to_y, lab_to_ify, sdiv, hdiv


SRGB_GAMMA, LAB_CBRT, SRGB_INV_GAMMA, LAB_TO_Y, LAB_TO_IFY, SDIV, HDIV180 = _tables()

_SRGB2XYZ = np.array([0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227])
_XYZ2SRGB = np.array([3.240479, -1.53715, -0.498535, -0.969256, 1.875991, 0.041556, 0.055648, -0.204043, 1.057311])
_D65 = np.array([0.950456, 1.0, 1.088754])


def _fwd_coeffs():
    scale = np.array([(1 << LAB_SHIFT) / _D65[0], float(1 << LAB_SHIFT), (1 << LAB_SHIFT) / _D65[2]])
    return _round_half_even(scale[:, None] * _SRGB2XYZ.reshape(3, 3)).astype(np.int64)   # rows X,Y,Z x cols R,G,B


def _inv_coeffs():
    # coeffs[ch][k] = round(2^12 * XYZ2sRGB[ch][k] * whitept[k])
    return _round_half_even((1 << LAB_SHIFT) * _XYZ2SRGB.reshape(3, 3) * _D65[None, :]).astype(np.int64)


FWD_C = _fwd_coeffs()
INV_C = _inv_coeffs()


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _cdiv(a, b):
    """C integer division (truncation toward zero) on int64 arrays."""
    q = np.abs(a) // b
    return np.where(a < 0, -q, q)


# ------------------------------------------------------------------------------------------
# colour conversions
# ------------------------------------------------------------------------------------------
def rgb2lab_u8(rgb: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(img, COLOR_RGB2LAB) for uint8 (RGB2Lab_b, sRGB gamma)."""
    R = SRGB_GAMMA[rgb[..., 0]].astype(np.int64)
    G = SRGB_GAMMA[rgb[..., 1]].astype(np.int64)
    B = SRGB_GAMMA[rgb[..., 2]].astype(np.int64)
    C = FWD_C
    fX = LAB_CBRT[_descale(R * C[0, 0] + G * C[0, 1] + B * C[0, 2], LAB_SHIFT)].astype(np.int64)
    fY = LAB_CBRT[_descale(R * C[1, 0] + G * C[1, 1] + B * C[1, 2], LAB_SHIFT)].astype(np.int64)
    fZ = LAB_CBRT[_descale(R * C[2, 0] + G * C[2, 1] + B * C[2, 2], LAB_SHIFT)].astype(np.int64)
    Lscale = (116 * 255 + 50) // 100
    Lshift = -((16 * 255 * (1 << LAB_SHIFT2) + 50) // 100)
    L = _descale(Lscale * fY + Lshift, LAB_SHIFT2)
    a = _descale(500 * (fX - fY) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2)
    b = _descale(200 * (fY - fZ) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2)
    return np.clip(np.stack([L, a, b], -1), 0, 255).astype(np.uint8)


def _ab_to_xz(v):
    """abToXZ_b table entry for index value v (computed, not tabulated)."""
    lo = _cdiv(v * 108, 841) - (LAB_BASE * 16 // 116 * 108 // 841)
    hi = _cdiv(_cdiv(v * v, LAB_BASE) * v, LAB_BASE)
    return np.where(v <= 3390, lo, hi)


def lab2rgb_u8(lab: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(img, COLOR_LAB2RGB) for uint8 (bit-exact integer path Lab2RGBinteger)."""
    LL = lab[..., 0].astype(np.int64)
    aa = lab[..., 1].astype(np.int64)
    bb = lab[..., 2].astype(np.int64)
    y = LAB_TO_Y[LL]
    ify = LAB_TO_IFY[LL]
    adiv = ((5 * aa * 53687 + (1 << 7)) >> 13) - 128 * LAB_BASE // 500
    bdiv = ((bb * 41943 + (1 << 4)) >> 9) - 128 * LAB_BASE // 200 + 1
    x = _ab_to_xz(ify + adiv)
    z = _ab_to_xz(ify - bdiv)
    shift = LAB_SHIFT + (14 - INV_GAMMA_SHIFT)
    C = INV_C
    out = []
    for ch in range(3):
        v = _descale(C[ch, 0] * x + C[ch, 1] * y + C[ch, 2] * z, shift)
        v = np.clip(v, 0, INV_GAMMA_TAB_SIZE - 1)
        out.append(SRGB_INV_GAMMA[v])
    return np.stack(out, -1).astype(np.uint8)


def rgb2hsv_u8(rgb: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(img, COLOR_RGB2HSV) for uint8 (H in [0,180))."""
    r = rgb[..., 0].astype(np.int64)
    g = rgb[..., 1].astype(np.int64)
    b = rgb[..., 2].astype(np.int64)
    v = np.maximum(np.maximum(r, g), b)
    vmin = np.minimum(np.minimum(r, g), b)
    diff = v - vmin
    s = (diff * SDIV[v] + (1 << 11)) >> 12
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * HDIV180[diff] + (1 << 11)) >> 12
    h = h + np.where(h < 0, 180, 0)
    return np.stack([np.clip(h, 0, 255), s, v], -1).astype(np.uint8)


_SECTOR = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])   # (b, g, r) <- tab idx


def hsv2rgb_u8(hsv: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(img, COLOR_HSV2RGB) for uint8: float32 sector formula, x255, round."""
    f32 = np.float32
    h = hsv[..., 0].astype(f32)
    s = hsv[..., 1].astype(f32) * f32(1.0 / 255.0)
    v = hsv[..., 2].astype(f32) * f32(1.0 / 255.0)
    h = h * f32(6.0 / 180.0)
    h = np.where(h >= f32(6.0), h - f32(6.0), h)      # u8 h <= 255 -> h*hscale < 8.5 : at most one wrap
    sector = np.floor(h).astype(np.int64)
    hf = (h - sector.astype(f32)).astype(f32)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    hf = np.where(bad, f32(0), hf)
    one = f32(1.0)
    tab = np.stack([v, v * (one - s), v * (one - s * hf), v * (one - s * (one - hf))], -1).astype(f32)
    idx = _SECTOR[sector]                                # (..., 3) in b,g,r order
    bgr = np.take_along_axis(tab, idx, axis=-1)
    gray = (hsv[..., 1] == 0)[..., None]
    bgr = np.where(gray, v[..., None], bgr).astype(f32)
    out = np.clip(np.rint(bgr * f32(255.0)), 0, 255).astype(np.uint8)
    return out[..., ::-1]                                # -> r,g,b


# ------------------------------------------------------------------------------------------
# CLAHE
# ------------------------------------------------------------------------------------------
def _reflect101(idx, n):
    """BORDER_REFLECT_101 index map (cv::borderInterpolate: reflect until inside, so images
    smaller than the border bounce more than once; a 1-pixel axis maps everything to 0)."""
    idx = np.abs(np.asarray(idx))
    if n == 1:
        return np.zeros_like(idx)
    period = 2 * (n - 1)
    idx = idx % period
    return np.where(idx >= n, period - idx, idx)


def clahe_u8(src: np.ndarray, clip_limit: float, grid: int) -> np.ndarray:
    """cv2.createCLAHE(clipLimit, (grid, grid)).apply(src) for a uint8 plane."""
    H, W = src.shape
    tx_n = ty_n = grid
    if W % tx_n == 0 and H % ty_n == 0:
        ext = src
    else:   # note: BOTH pads are applied, a dimension that divides evenly gets a full extra `grid`
        ph, pw = ty_n - (H % ty_n), tx_n - (W % tx_n)
        ys = _reflect101(np.arange(H + ph), H)
        xs = _reflect101(np.arange(W + pw), W)
        ext = src[np.ix_(ys, xs)]
    th, tw = ext.shape[0] // ty_n, ext.shape[1] // tx_n
    area = th * tw
    lut_scale = np.float32(255.0) / np.float32(area)
    clip = 0
    if clip_limit > 0.0:
        clip = max(int(clip_limit * area / 256), 1)
    luts = np.zeros((ty_n, tx_n, 256), np.uint8)
    for ty in range(ty_n):
        for tx in range(tx_n):
            tile = ext[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
            hist = np.bincount(tile.ravel(), minlength=256).astype(np.int64)
            if clip > 0:
                clipped = int(np.maximum(hist - clip, 0).sum())
                hist = np.minimum(hist, clip)
                batch = clipped // 256
                residual = clipped - batch * 256
                hist += batch
                if residual != 0:
                    step = max(256 // residual, 1)
                    i = 0
                    while i < 256 and residual > 0:
                        hist[i] += 1
                        i += step
                        residual -= 1
            cdf = np.cumsum(hist)
            luts[ty, tx] = np.clip(np.rint(cdf.astype(np.float32) * lut_scale), 0, 255).astype(np.uint8)
    f32 = np.float32
    inv_tw, inv_th = f32(1.0) / f32(tw), f32(1.0) / f32(th)
    txf = np.arange(W, dtype=f32) * inv_tw - f32(0.5)
    tx1 = np.floor(txf).astype(np.int64)
    xa = (txf - tx1.astype(f32)).astype(f32)
    xa1 = (f32(1.0) - xa).astype(f32)
    tx2 = np.minimum(tx1 + 1, tx_n - 1)
    tx1 = np.maximum(tx1, 0)
    tyf = np.arange(H, dtype=f32) * inv_th - f32(0.5)
    ty1 = np.floor(tyf).astype(np.int64)
    ya = (tyf - ty1.astype(f32)).astype(f32)
    ya1 = (f32(1.0) - ya).astype(f32)
    ty2 = np.minimum(ty1 + 1, ty_n - 1)
    ty1 = np.maximum(ty1, 0)
    s = src.astype(np.int64)
    Y1, X1 = ty1[:, None], tx1[None, :]
    Y2, X2 = ty2[:, None], tx2[None, :]
    l11 = luts[Y1, X1, s].astype(f32)
    l12 = luts[Y1, X2, s].astype(f32)
    l21 = luts[Y2, X1, s].astype(f32)
    l22 = luts[Y2, X2, s].astype(f32)
    XA, XA1, YA, YA1 = xa[None, :], xa1[None, :], ya[:, None], ya1[:, None]
    res = (l11 * XA1 + l12 * XA) * YA1 + (l21 * XA1 + l22 * XA) * YA
    return np.clip(np.rint(res.astype(f32)), 0, 255).astype(np.uint8)


# ------------------------------------------------------------------------------------------
# Gaussian blur (8-bit fixed point) and addWeighted
# ------------------------------------------------------------------------------------------
def gaussian_kernel_q8(sigma: float):
    """ksize = cvRound(sigma*6+1)|1 ; 8.8 fixed-point taps by error-diffusion rounding, sum 256."""
    n = int(np.rint(sigma * 6 + 1)) | 1
    x = np.arange(n, dtype=np.float64) - (n - 1) / 2
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    k = k / k.sum()
    taps = np.zeros(n, np.int64)
    err = 0.0
    tot = 0
    for i in range(n // 2):
        adj = k[i] * 256.0 + err
        v0 = int(np.rint(adj))
        err = adj - v0
        taps[i] = taps[n - 1 - i] = v0
        tot += 2 * v0
    taps[n // 2] = 256 - tot
    return taps


def gaussian_blur_u8(img: np.ndarray, sigma: float) -> np.ndarray:
    """cv2.GaussianBlur(img, (0,0), sigma) for uint8, BORDER_REFLECT_101, exact integer math:
    rows: sum kx*src (8.8), columns: sum ky*row (16.16), (v + 2^15) >> 16."""
    taps = gaussian_kernel_q8(sigma)
    r = len(taps) // 2
    H, W = img.shape[:2]
    ys = _reflect101(np.arange(-r, H + r), H)
    xs = _reflect101(np.arange(-r, W + r), W)
    p = img[np.ix_(ys, xs)].astype(np.int64)
    hor = np.zeros((H + 2 * r, W) + img.shape[2:], np.int64)
    for i, t in enumerate(taps):
        hor += t * p[:, i:i + W]
    ver = np.zeros((H, W) + img.shape[2:], np.int64)
    for i, t in enumerate(taps):
        ver += t * hor[i:i + H]
    return np.clip((ver + (1 << 15)) >> 16, 0, 255).astype(np.uint8)


def add_weighted_u8(a: np.ndarray, alpha: float, b: np.ndarray, beta: float) -> np.ndarray:
    """cv2.addWeighted(a, alpha, b, beta, 0) for uint8: float32 a*alpha + b*beta, round, saturate."""
    f32 = np.float32
    r = a.astype(f32) * f32(alpha) + b.astype(f32) * f32(beta)
    return np.clip(np.rint(r), 0, 255).astype(np.uint8)


# ------------------------------------------------------------------------------------------
# the three stages and the two reference pipelines
# ------------------------------------------------------------------------------------------
def local_contrast(img: np.ndarray, clip_limit: float, grid: int) -> np.ndarray:
    """wow_sr.py:190-193 / farm_sr.py:79-86."""
    lab = rgb2lab_u8(img)
    lab[..., 0] = clahe_u8(lab[..., 0], clip_limit, grid)
    return lab2rgb_u8(lab)


def unsharp(img: np.ndarray, sigma: float, w_img: float, w_blur: float) -> np.ndarray:
    """wow_sr.py:196-197 / farm_sr.py:66-69."""
    return add_weighted_u8(img, w_img, gaussian_blur_u8(img, sigma), w_blur)


def vegetation(img: np.ndarray, gain: float, hue_lo: int = 35, hue_hi: int = 85) -> np.ndarray:
    """wow_sr.py:200-207 / farm_sr.py:94-106: HSV float32, S*gain (f32) clipped where
    hue_lo < H < hue_hi, astype(uint8) truncation, back to RGB."""
    hsv = rgb2hsv_u8(img).astype(np.float32)
    mask = (hsv[..., 0] > hue_lo) & (hsv[..., 0] < hue_hi)
    boosted = np.clip(hsv[..., 1] * np.float32(gain), 0, 255)
    hsv[..., 1] = np.where(mask, boosted, hsv[..., 1])
    return hsv2rgb_u8(hsv.astype(np.uint8))


def enhance_for_crops(img: np.ndarray) -> np.ndarray:
    """`_enhance_for_crops` (wow_sr.py:187-209)."""
    out = local_contrast(img, 2.5, 8)
    out = unsharp(out, 1.2, 1.4, -0.4)
    return vegetation(out, 1.2)


def farm_postprocess(img: np.ndarray) -> np.ndarray:
    """apply_farm_sr steps 2-4 (farm_sr.py:170-178): CLAHE 2.5/8, unsharp strength 1.2 radius 1.5, x1.3."""
    out = local_contrast(img, 2.5, 8)
    out = unsharp(out, 1.5, 1.0 + 1.2, -1.2)
    return vegetation(out, 1.3)


def postprocess(img: np.ndarray, clahe_clip, clahe_grid, blur_sigma, w_img, w_blur, hue_lo, hue_hi, sat_gain,
                stages: int) -> np.ndarray:
    out = img
    if stages & 1:
        out = local_contrast(out, clahe_clip, clahe_grid)
    if stages & 2:
        out = unsharp(out, blur_sigma, w_img, w_blur)
    if stages & 4:
        out = vegetation(out, sat_gain, hue_lo, hue_hi)
    return out
