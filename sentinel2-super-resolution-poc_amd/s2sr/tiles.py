"""Host-side planning of the XYZ tile pyramid (the step after the SR path; reference
server/app/tiling.py:102-186 = gdalwarp to EPSG:3857 + gdal2tiles.py --xyz --resampling average).

The geometry is resolved here in float64 into plain integer / float32 tables; the HIP kernels
(csrc/tiles.hip) then do pixel arithmetic only:

  warp      output pixel -> source pixel coordinates, sampled on a node grid every WARP_STEP output
            pixels (the projection is smooth: linear interpolation between nodes is exact to
            << 0.01 px), bilinear sampling of the source on the device;
  base      tiles of the deepest zoom: every tile pixel averages the source pixels whose centres
            fall inside its footprint (nearest source pixel when the footprint holds none), given
            as [lo, hi] column / row index tables per tile column / row;
  overview  every shallower zoom from the four children of a tile: mean of the valid (alpha > 0)
            pixels of each 2x2 group.

Parity note: GDAL is not available to this build, so these are this build's definitions of
"bilinear" and "average" (DESIGN.md section 7), not bit-for-bit gdalwarp / gdal2tiles.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

from . import geo

WARP_STEP = 64      # output pixels between nodes.  The projection pair is smooth on this scale: against projecting every pixel the
                    # interpolated source coordinate is off by < 1e-3 px at 64 (tests/test_tiles_cpu.py), and the plan of a 4096 x 4096
                    # raster costs 0.6 ms instead of 9 (r04: 16 -> 66k nodes through the Krueger series per pyramid)


@dataclass
class WarpPlan:
    out_h: int
    out_w: int
    placement: geo.Placement          # of the output raster, EPSG:3857
    grid: np.ndarray                  # float32 [gh, gw, 2]: source (col, row) in pixel-centre coordinates at the nodes
    step: int = WARP_STEP


def plan_warp(width: int, height: int, src: geo.Placement, crs: geo.CRS) -> WarpPlan:
    """Output grid like GDAL's suggested warp output: the extent of the densified source outline,
    square pixels sized so that the pixel count along the diagonal is preserved."""
    t = np.linspace(0.0, 1.0, 21)
    w, s, e, n = src.bounds(width, height)
    ex = np.concatenate([w + (e - w) * t, np.full(21, e), e - (e - w) * t, np.full(21, w)])
    ey = np.concatenate([np.full(21, n), n - (n - s) * t, np.full(21, s), s + (n - s) * t])
    lon, lat = crs.to_lonlat(ex, ey)
    mx, my = geo.lonlat_to_mercator(lon, lat)
    mw, me, ms, mn = float(mx.min()), float(mx.max()), float(my.min()), float(my.max())
    res = math.hypot(me - mw, mn - ms) / math.hypot(width, height)
    ow, oh = max(1, int((me - mw) / res + 0.5)), max(1, int((mn - ms) / res + 0.5))
    dst = geo.Placement(mw, mn, res, res)
    gh, gw = (oh - 1 + WARP_STEP - 1) // WARP_STEP + 1, (ow - 1 + WARP_STEP - 1) // WARP_STEP + 1
    jj, ii = np.meshgrid(np.arange(gw, dtype=np.float64) * WARP_STEP, np.arange(gh, dtype=np.float64) * WARP_STEP)
    X = mw + (jj + 0.5) * res
    Y = mn - (ii + 0.5) * res
    lon, lat = geo.mercator_to_lonlat(X, Y)
    sx, sy = crs.from_lonlat(lon, lat)
    u = (sx - src.x0) / src.dx - 0.5
    v = (src.y0 - sy) / src.dy - 0.5
    return WarpPlan(oh, ow, dst, np.stack([u, v], -1).astype(np.float32))


@dataclass
class LevelPlan:
    zoom: int
    tminx: int
    tminy: int
    tmaxx: int
    tmaxy: int                        # TMS numbering; arrays are stored north row first

    @property
    def nx(self) -> int:
        return self.tmaxx - self.tminx + 1

    @property
    def ny(self) -> int:
        return self.tmaxy - self.tminy + 1


def plan_levels(bounds, min_zoom: int, max_zoom: int) -> List[LevelPlan]:
    """Deepest zoom first."""
    return [LevelPlan(z, *geo.tile_range(bounds, z)) for z in range(max_zoom, min_zoom - 1, -1)]


def _footprint(a: np.ndarray, b: np.ndarray, n: int):
    """Index ranges [lo, hi] of the source pixels whose centres (index + 0.5) lie in [a, b), a and b in
    continuous source-pixel coordinates (columns rightwards, rows downwards).  A footprint that holds
    no centre falls back to the pixel containing its midpoint.  Clipped to the raster: lo > hi = nothing."""
    lo = np.ceil(a - 0.5).astype(np.int64)
    hi = np.ceil(b - 0.5).astype(np.int64) - 1
    centre = np.floor((a + b) * 0.5).astype(np.int64)
    empty = hi < lo
    lo = np.where(empty, centre, lo)
    hi = np.where(empty, centre, hi)
    return np.maximum(lo, 0).astype(np.int32), np.minimum(hi, n - 1).astype(np.int32)


def plan_base(level: LevelPlan, src: geo.Placement, width: int, height: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """-> col_lo, col_hi [nx*256], row_lo, row_hi [ny*256] (rows north to south)."""
    r = geo.resolution(level.zoom)
    px = np.arange(level.nx * geo.TILE, dtype=np.float64)
    x_lo = level.tminx * geo.TILE * r - geo.ORIGIN_SHIFT + px * r          # west edge of every tile-pixel column
    col_lo, col_hi = _footprint((x_lo - src.x0) / src.dx, (x_lo + r - src.x0) / src.dx, width)
    py = np.arange(level.ny * geo.TILE, dtype=np.float64)
    y_hi = (level.tmaxy + 1) * geo.TILE * r - geo.ORIGIN_SHIFT - py * r    # north edge of every tile-pixel row
    row_lo, row_hi = _footprint((src.y0 - y_hi) / src.dy, (src.y0 - (y_hi - r)) / src.dy, height)
    return col_lo, col_hi, row_lo, row_hi


def overview_offsets(parent: LevelPlan, child: LevelPlan) -> Tuple[int, int]:
    """Child-array (col, row) of the north-west child of the parent array's first tile."""
    return 2 * parent.tminx - child.tminx, child.tmaxy - (2 * parent.tmaxy + 1)
