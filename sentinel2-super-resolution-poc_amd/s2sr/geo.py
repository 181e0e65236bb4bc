"""Geodesy and tile arithmetic for the XYZ pyramid that follows the SR path (reference
server/app/tiling.py:102-186 shells out to gdalwarp -t_srs EPSG:3857 and gdal2tiles.py --xyz;
neither GDAL nor PROJ is required here).

* WGS84 transverse Mercator (UTM zones, EPSG:326zz / 327zz) by the Krueger n-series (Karney 2011,
  6th order: sub-micrometre inside a zone), spherical Web Mercator (EPSG:3857), plain EPSG:4326;
* the GeoTIFF keys that name the CRS (GeoKeyDirectory 34735: ProjectedCSTypeGeoKey 3072,
  GeographicTypeGeoKey 2048) and the tiepoint / pixel-scale placement;
* the global Web-Mercator tile scheme of gdal2tiles (256-pixel tiles, TMS row order internally,
  XYZ row = 2^z - 1 - TMS row).
Everything is float64 numpy on the host; the device only sees pixel-space coordinates.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

WGS84_A = 6378137.0
WGS84_F = 1.0 / 298.257223563
ORIGIN_SHIFT = math.pi * WGS84_A           # 20037508.342789244
TILE = 256


# ---------------------------------------------------------------------------------------------
# Web Mercator
# ---------------------------------------------------------------------------------------------
def lonlat_to_mercator(lon, lat):
    lon, lat = np.asarray(lon, np.float64), np.asarray(lat, np.float64)
    x = np.radians(lon) * WGS84_A
    y = np.log(np.tan(np.pi / 4.0 + np.radians(lat) / 2.0)) * WGS84_A
    return x, y


def mercator_to_lonlat(x, y):
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    lon = np.degrees(x / WGS84_A)
    lat = np.degrees(2.0 * np.arctan(np.exp(y / WGS84_A)) - np.pi / 2.0)
    return lon, lat


# ---------------------------------------------------------------------------------------------
# Transverse Mercator (Krueger series)
# ---------------------------------------------------------------------------------------------
_N = WGS84_F / (2.0 - WGS84_F)
_A = WGS84_A / (1.0 + _N) * (1.0 + _N ** 2 / 4.0 + _N ** 4 / 64.0 + _N ** 6 / 256.0)
_ALPHA = (
    _N / 2 - 2 * _N ** 2 / 3 + 5 * _N ** 3 / 16 + 41 * _N ** 4 / 180 - 127 * _N ** 5 / 288 + 7891 * _N ** 6 / 37800,
    13 * _N ** 2 / 48 - 3 * _N ** 3 / 5 + 557 * _N ** 4 / 1440 + 281 * _N ** 5 / 630 - 1983433 * _N ** 6 / 1935360,
    61 * _N ** 3 / 240 - 103 * _N ** 4 / 140 + 15061 * _N ** 5 / 26880 + 167603 * _N ** 6 / 181440,
    49561 * _N ** 4 / 161280 - 179 * _N ** 5 / 168 + 6601661 * _N ** 6 / 7257600,
    34729 * _N ** 5 / 80640 - 3418889 * _N ** 6 / 1995840,
    212378941 * _N ** 6 / 319334400,
)
_BETA = (
    _N / 2 - 2 * _N ** 2 / 3 + 37 * _N ** 3 / 96 - _N ** 4 / 360 - 81 * _N ** 5 / 512 + 96199 * _N ** 6 / 604800,
    _N ** 2 / 48 + _N ** 3 / 15 - 437 * _N ** 4 / 1440 + 46 * _N ** 5 / 105 - 1118711 * _N ** 6 / 3870720,
    17 * _N ** 3 / 480 - 37 * _N ** 4 / 840 - 209 * _N ** 5 / 4480 + 5569 * _N ** 6 / 90720,
    4397 * _N ** 4 / 161280 - 11 * _N ** 5 / 504 - 830251 * _N ** 6 / 7257600,
    4583 * _N ** 5 / 161280 - 108847 * _N ** 6 / 3991680,
    20648693 * _N ** 6 / 638668800,
)
_E = math.sqrt(WGS84_F * (2.0 - WGS84_F))
K0, FALSE_E, FALSE_N_SOUTH = 0.9996, 500000.0, 10000000.0


def tm_forward(lon, lat, lon0, south=False):
    """geodetic degrees -> UTM easting, northing (metres) on central meridian lon0."""
    lam = np.radians(np.asarray(lon, np.float64) - lon0)
    phi = np.radians(np.asarray(lat, np.float64))
    s = np.sin(phi)
    t = np.sinh(np.arctanh(s) - _E * np.arctanh(_E * s))          # tan of the conformal latitude
    xi = np.arctan2(t, np.cos(lam))
    eta = np.arctanh(np.sin(lam) / np.sqrt(1.0 + t * t))
    x, y = eta.copy(), xi.copy()
    for j, a in enumerate(_ALPHA, start=1):
        y += a * np.sin(2 * j * xi) * np.cosh(2 * j * eta)
        x += a * np.cos(2 * j * xi) * np.sinh(2 * j * eta)
    return FALSE_E + K0 * _A * x, (FALSE_N_SOUTH if south else 0.0) + K0 * _A * y


def tm_inverse(e, n, lon0, south=False):
    """UTM easting, northing -> geodetic degrees."""
    xi = (np.asarray(n, np.float64) - (FALSE_N_SOUTH if south else 0.0)) / (K0 * _A)
    eta = (np.asarray(e, np.float64) - FALSE_E) / (K0 * _A)
    xi0, eta0 = xi.copy(), eta.copy()
    for j, b in enumerate(_BETA, start=1):
        xi0 -= b * np.sin(2 * j * xi) * np.cosh(2 * j * eta)
        eta0 -= b * np.cos(2 * j * xi) * np.sinh(2 * j * eta)
    tau = np.sin(xi0) / np.sqrt(np.sinh(eta0) ** 2 + np.cos(xi0) ** 2)   # tan of the conformal latitude
    t = tau.copy()                                                    # Newton on tan(phi)
    for _ in range(6):
        sig = np.sinh(_E * np.arctanh(_E * t / np.sqrt(1.0 + t * t)))
        f = t * np.sqrt(1.0 + sig * sig) - sig * np.sqrt(1.0 + t * t) - tau
        df = (np.sqrt((1.0 + sig * sig) * (1.0 + t * t)) - sig * t) * (1.0 - _E * _E) * np.sqrt(1.0 + t * t) / \
             (1.0 + (1.0 - _E * _E) * t * t)
        t = t - f / df
    lat = np.degrees(np.arctan(t))
    lon = lon0 + np.degrees(np.arctan2(np.sinh(eta0), np.cos(xi0)))
    return lon, lat


# ---------------------------------------------------------------------------------------------
# CRS handling
# ---------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class CRS:
    epsg: int

    @property
    def kind(self) -> str:
        if self.epsg == 3857:
            return "mercator"
        if self.epsg == 4326:
            return "geographic"
        if 32601 <= self.epsg <= 32660 or 32701 <= self.epsg <= 32760:
            return "utm"
        raise ValueError(f"EPSG:{self.epsg} is not supported (UTM 326xx/327xx, 3857 and 4326 are)")

    def to_lonlat(self, x, y):
        k = self.kind
        if k == "geographic":
            return np.asarray(x, np.float64), np.asarray(y, np.float64)
        if k == "mercator":
            return mercator_to_lonlat(x, y)
        zone = self.epsg % 100
        return tm_inverse(x, y, zone * 6 - 183, south=self.epsg >= 32700)

    def from_lonlat(self, lon, lat):
        k = self.kind
        if k == "geographic":
            return np.asarray(lon, np.float64), np.asarray(lat, np.float64)
        if k == "mercator":
            return lonlat_to_mercator(lon, lat)
        zone = self.epsg % 100
        return tm_forward(lon, lat, zone * 6 - 183, south=self.epsg >= 32700)

    def __str__(self):
        return f"EPSG:{self.epsg}"


def epsg_from_geokeys(geokeys) -> Optional[int]:
    """GeoKeyDirectoryTag (34735): header of 4 shorts, then (key, location, count, value) entries."""
    if not geokeys or len(geokeys) < 4:
        return None
    keys = {}
    for i in range(int(geokeys[3])):
        k, loc, _cnt, val = (int(v) for v in geokeys[4 + 4 * i: 8 + 4 * i])
        if loc == 0:
            keys[k] = val
    for k in (3072, 2048):            # ProjectedCSTypeGeoKey, then GeographicTypeGeoKey
        if k in keys and 0 < keys[k] < 32767:
            return keys[k]
    return None


@dataclass(frozen=True)
class Placement:
    """North-up placement of a raster: x = x0 + col*dx, y = y0 - row*dy (dx, dy > 0), pixel edges."""
    x0: float
    y0: float
    dx: float
    dy: float

    def bounds(self, width: int, height: int) -> Tuple[float, float, float, float]:
        return self.x0, self.y0 - height * self.dy, self.x0 + width * self.dx, self.y0   # west, south, east, north


def placement_from_tags(tags) -> Optional[Placement]:
    """ModelTiepointTag (33922) + ModelPixelScaleTag (33550), or an axis-aligned ModelTransformationTag."""
    if 33550 in tags and 33922 in tags:
        sx, sy = float(tags[33550][0]), float(tags[33550][1])
        i, j, _k, x, y, _z = (float(v) for v in tags[33922][:6])
        return Placement(x - i * sx, y + j * sy, sx, sy)
    if 34264 in tags:
        m = [float(v) for v in tags[34264]]
        if m[1] == 0.0 and m[4] == 0.0 and m[0] > 0 and m[5] < 0:
            return Placement(m[3], m[7], m[0], -m[5])
    return None


# ---------------------------------------------------------------------------------------------
# global Web-Mercator tiles (TMS rows internally, like gdal2tiles' GlobalMercator)
# ---------------------------------------------------------------------------------------------
def resolution(zoom: int) -> float:
    return 2.0 * ORIGIN_SHIFT / (TILE * 2 ** zoom)


def meters_to_tile(mx: float, my: float, zoom: int) -> Tuple[int, int]:
    res = resolution(zoom)
    px, py = (mx + ORIGIN_SHIFT) / res, (my + ORIGIN_SHIFT) / res
    return int(math.ceil(px / TILE) - 1), int(math.ceil(py / TILE) - 1)


def tile_bounds(tx: int, ty: int, zoom: int) -> Tuple[float, float, float, float]:
    res = resolution(zoom)
    return (tx * TILE * res - ORIGIN_SHIFT, ty * TILE * res - ORIGIN_SHIFT,
            (tx + 1) * TILE * res - ORIGIN_SHIFT, (ty + 1) * TILE * res - ORIGIN_SHIFT)


def tile_range(bounds, zoom: int) -> Tuple[int, int, int, int]:
    """(tminx, tminy, tmaxx, tmaxy) in TMS numbering covering west, south, east, north (metres)."""
    w, s, e, n = bounds
    tminx, tminy = meters_to_tile(w, s, zoom)
    tmaxx, tmaxy = meters_to_tile(e, n, zoom)
    lim = 2 ** zoom - 1
    return max(0, tminx), max(0, tminy), min(lim, tmaxx), min(lim, tmaxy)


def xyz_row(ty: int, zoom: int) -> int:
    return (2 ** zoom - 1) - ty
