"""ORACLE (test infrastructure only -- never imported by the product path).

numpy restatement of this build's XYZ tile pyramid (sentinel2-super-resolution-poc_amd/csrc/tiles.hip
+ s2sr/tiles.py), the step the reference runs right after the SR path by shelling out to GDAL
(server/app/tiling.py:102-186: `gdalwarp -t_srs EPSG:3857 -r bilinear`, then
`gdal2tiles.py --zoom a-b --tilesize 256 --resampling average --xyz`).

PARITY UNPINNED against the reference: the arithmetic lives in GDAL (gdal-bin, no version pin in the
reference's Dockerfile / requirements, not present in this image) and the reference holds no tile
fixture.  This file therefore pins the build's OWN definitions, derived independently of the
planner tables in s2sr/tiles.py:
  * warp: node grid of source coordinates, linear interpolation between nodes, 4-tap bilinear with
    edge replication in float32 (single operations, fixed order), alpha = inside the source;
  * base tile pixel: rounded mean of the valid source pixels whose CENTRES lie inside the tile
    pixel's Web-Mercator footprint [west, east) x (south, north], nearest pixel when none does;
  * overview pixel: rounded mean of the valid pixels of the 2x2 group below it.
The geodesy itself (UTM <-> WGS84 <-> EPSG:3857, tile numbering) is checked against published
values in tests/test_tiles_cpu.py.
"""
from __future__ import annotations

import math

import numpy as np

R = 6378137.0
SHIFT = math.pi * R
f32 = np.float32


def warp_bilinear(rgb: np.ndarray, grid: np.ndarray, step: int, out_h: int, out_w: int) -> np.ndarray:
    H, W = rgb.shape[:2]
    gh, gw = grid.shape[:2]
    oy, ox = np.mgrid[0:out_h, 0:out_w]
    gi, gj = oy // step, ox // step
    gi1, gj1 = np.minimum(gi + 1, gh - 1), np.minimum(gj + 1, gw - 1)
    fi = ((oy - gi * step).astype(f32) * f32(1.0 / step)).astype(f32)
    fj = ((ox - gj * step).astype(f32) * f32(1.0 / step)).astype(f32)
    g = grid.astype(f32)

    def interp(k):
        a, b, c, d = g[gi, gj, k], g[gi, gj1, k], g[gi1, gj, k], g[gi1, gj1, k]
        t0 = (a + ((b - a) * fj).astype(f32)).astype(f32)
        t1 = (c + ((d - c) * fj).astype(f32)).astype(f32)
        return (t0 + ((t1 - t0) * fi).astype(f32)).astype(f32)

    u, v = interp(0), interp(1)
    valid = (u >= f32(-0.5)) & (u <= f32(W) - f32(0.5)) & (v >= f32(-0.5)) & (v <= f32(H) - f32(0.5))
    xf, yf = np.floor(u), np.floor(v)
    fx, fy = (u - xf).astype(f32), (v - yf).astype(f32)
    xi, yi = np.where(valid, xf, 0).astype(np.int64), np.where(valid, yf, 0).astype(np.int64)
    x0, x1 = np.clip(xi, 0, W - 1), np.clip(xi + 1, 0, W - 1)
    y0, y1 = np.clip(yi, 0, H - 1), np.clip(yi + 1, 0, H - 1)
    out = np.zeros((out_h, out_w, 4), np.uint8)
    src = rgb.astype(f32)
    for k in range(3):
        p00, p10, p01, p11 = src[y0, x0, k], src[y0, x1, k], src[y1, x0, k], src[y1, x1, k]
        t = (p00 + ((p10 - p00) * fx).astype(f32)).astype(f32)
        b = (p01 + ((p11 - p01) * fx).astype(f32)).astype(f32)
        val = (t + ((b - t) * fy).astype(f32)).astype(f32)
        out[..., k] = np.where(valid, (val + f32(0.5)).astype(f32).astype(np.int64), 0)
    out[..., 3] = np.where(valid, 255, 0)
    return out


def _mean_valid(px: np.ndarray) -> np.ndarray:
    """px [..., n, 4] -> [..., 4] rounded mean over entries with alpha > 0."""
    ok = px[..., 3] > 0
    n = ok.sum(-1)
    s = (px[..., :3].astype(np.int64) * ok[..., None]).sum(-2)
    out = np.zeros(px.shape[:-2] + (4,), np.uint8)
    nz = n > 0
    out[nz, :3] = ((s[nz] + (n[nz] // 2)[:, None]) // n[nz][:, None]).astype(np.uint8)
    out[nz, 3] = 255
    return out


def base_tile(rgba: np.ndarray, x0: float, y0: float, dx: float, dy: float, tx: int, ty: int, zoom: int) -> np.ndarray:
    """One 256x256 RGBA tile (TMS numbering) straight from the geometry, pixel by pixel."""
    H, W = rgba.shape[:2]
    res = 2.0 * SHIFT / (256 * 2 ** zoom)
    west, north = tx * 256 * res - SHIFT, (ty + 1) * 256 * res - SHIFT
    cx = x0 + (np.arange(W) + 0.5) * dx          # centres of the source pixels, metres
    cy = y0 - (np.arange(H) + 0.5) * dy
    out = np.zeros((256, 256, 4), np.uint8)
    for py in range(256):
        n_edge, s_edge = north - py * res, north - (py + 1) * res
        rows = np.nonzero((cy <= n_edge) & (cy > s_edge))[0]
        if rows.size == 0:
            r = int(math.floor((y0 - 0.5 * (n_edge + s_edge)) / dy))
            rows = np.array([r]) if 0 <= r < H else rows
        for px in range(256):
            w_edge, e_edge = west + px * res, west + (px + 1) * res
            cols = np.nonzero((cx >= w_edge) & (cx < e_edge))[0]
            if cols.size == 0:
                c = int(math.floor((0.5 * (w_edge + e_edge) - x0) / dx))
                cols = np.array([c]) if 0 <= c < W else cols
            if rows.size and cols.size:
                blk = rgba[np.ix_(rows, cols)].reshape(1, -1, 4)
                out[py, px] = _mean_valid(blk)[0]
    return out


def overview(children: np.ndarray, ox: int, oy: int, pnx: int, pny: int) -> np.ndarray:
    """children [cny, cnx, 256, 256, 4] -> parents [pny, pnx, 256, 256, 4]."""
    cny, cnx = children.shape[:2]
    mosaic = np.zeros(((2 * pny) * 256, (2 * pnx) * 256, 4), np.uint8)
    for j in range(2 * pny):
        for i in range(2 * pnx):
            cy, cx = oy + j, ox + i
            if 0 <= cy < cny and 0 <= cx < cnx:
                mosaic[j * 256:(j + 1) * 256, i * 256:(i + 1) * 256] = children[cy, cx]
    g = mosaic.reshape(pny * 256, 2, pnx * 256, 2, 4).transpose(0, 2, 1, 3, 4).reshape(pny * 256, pnx * 256, 4, 4)
    m = _mean_valid(g)
    return m.reshape(pny, 256, pnx, 256, 4).transpose(0, 2, 1, 3, 4).copy()
