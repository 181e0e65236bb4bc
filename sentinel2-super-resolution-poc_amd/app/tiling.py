"""Reprojection to Web Mercator and XYZ tile generation on the GPU: the same module surface as the
reference's GDAL helpers (reference server/app/tiling.py: RasterInfo :14-25, get_raster_info :28-99,
reproject_to_web_mercator :102-135, generate_xyz_tiles :138-186, create_tileset_metadata :189-224,
process_raster_to_tiles :227-275), without gdalinfo / gdalwarp / gdal2tiles.py subprocesses.

Scope: 8-bit RGB rasters (what the SR path writes) in UTM (EPSG:326xx / 327xx), EPSG:4326 or
EPSG:3857, north-up.  Resampling definitions are this build's (s2sr/tiles.py, csrc/tiles.hip);
GDAL is not available to compare against, see DESIGN.md.
"""
from __future__ import annotations

import json
import logging
import struct
import threading
import zlib
from dataclasses import dataclass
from pathlib import Path
from typing import Optional

import numpy as np
from PIL import Image

from s2sr import geo, native, tiles
from s2sr import rasterio_lite as rio
from s2sr import tiff_lite

logger = logging.getLogger("tiling")

_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def filter_sub_rgba(tiles_arr: np.ndarray) -> np.ndarray:
    """[..., 256, 256, 4] uint8 -> [..., 256, 1 + 1024]: every row behind its PNG filter byte (type 1, Sub, bpp = 4; wraps modulo
    256).  Vectorised over the leading axes: the pyramid filters 8 tiles per call inside the encoder threads (per tile the numpy
    calls are interpreter overhead; a whole z18 level in one call is 3 GB through one thread, 3x slower than either)."""
    lead, (h, w, _) = tiles_arr.shape[:-3], tiles_arr.shape[-3:]
    rows = tiles_arr.reshape(lead + (h, w * 4))
    raw = np.empty(lead + (h, w * 4 + 1), np.uint8)
    raw[..., 0] = 1
    raw[..., 1:5] = rows[..., :4]
    np.subtract(rows[..., 4:], rows[..., :-4], out=raw[..., 5:])
    return raw


def _png_from_filtered(raw: np.ndarray, level: int, strategy: int) -> bytes:
    h, w = raw.shape[0], (raw.shape[1] - 1) // 4

    def chunk(kind: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)

    co = zlib.compressobj(level, zlib.DEFLATED, 15, 9, strategy)
    return (_PNG_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
            chunk(b"IDAT", co.compress(raw.tobytes()) + co.flush()) + chunk(b"IEND", b""))


def encode_png_rgba(tile: np.ndarray, level: int = 1, strategy: int = zlib.Z_RLE) -> bytes:
    """256x256x4 uint8 -> PNG bytes (8-bit RGBA, Sub filter on every row, one IDAT).  The defaults (level 1 + Z_RLE, the fast
    end of deflate, what cv2.imwrite uses) go through the native encoder (csrc/pngenc.hip, ~3x zlib on tile data, GIL released:
    tiles are encoded on a thread pool; gdal2tiles uses --processes 4 for this); other settings through zlib."""
    if level == 1 and strategy == zlib.Z_RLE:
        return native.png_encode(tile)
    return _png_from_filtered(filter_sub_rgba(tile), level, strategy)

_ENGINES: dict = {}                 # device index -> (engine, lock of its pyramid chain)
_ENGINES_LOCK = threading.Lock()


def _engine_and_lock():
    """The pyramid engine of the GPU this job was admitted to (app.sr_routes.GpuAdmission sets the thread's device; default:
    LOCAL_RANK, as everywhere in app.*) and the lock that keeps one pyramid chain at a time on it.  A weightless handle is enough
    for the pyramid kernels (raises without a gfx950 GPU)."""
    from app.cnn_super_resolution import current_device_index
    dev = current_device_index()
    with _ENGINES_LOCK:
        if dev not in _ENGINES:
            _ENGINES[dev] = (native.Engine(num_block=1, device=dev), threading.RLock())
        return _ENGINES[dev]


def _engine() -> native.Engine:
    return _engine_and_lock()[0]


@dataclass
class RasterInfo:
    """Information about a raster file (field for field the reference's dataclass)."""
    path: Path
    crs: str
    bounds: list          # [west, south, east, north] in native CRS
    bounds_4326: list     # [west, south, east, north] in EPSG:4326
    width: int
    height: int
    bands: int
    dtype: str


_GDAL_TYPE = {"uint8": "Byte", "uint16": "UInt16", "int16": "Int16", "uint32": "UInt32", "int32": "Int32",
              "float32": "Float32", "float64": "Float64"}


def _read(path: Path):
    # a raster this process wrote a moment ago and nobody touched since (a job's sr_tif, main.py:347-359) comes from memory
    hit = rio.recall_written(path)
    arr, tags = hit if hit is not None else tiff_lite.read_tiff(path)
    place = geo.placement_from_tags(tags)
    if place is None:
        raise ValueError(f"{path}: no north-up georeferencing (tiepoint + pixel scale) in the GeoTIFF tags")
    epsg = geo.epsg_from_geokeys(tags.get(rio.TAG_GEOKEYS))
    return arr, tags, place, geo.CRS(epsg if epsg else 4326)      # the reference defaults to EPSG:4326 too (:49)


def get_raster_info(raster_path: Path) -> RasterInfo:
    raster_path = Path(raster_path)
    arr, _tags, place, crs = _read(raster_path)
    h, w, b = arr.shape
    west, south, east, north = place.bounds(w, h)
    t = np.linspace(0.0, 1.0, 21)
    ex = np.concatenate([west + (east - west) * t, np.full(21, east), east - (east - west) * t, np.full(21, west)])
    ey = np.concatenate([np.full(21, north), north - (north - south) * t, np.full(21, south), south + (north - south) * t])
    lon, lat = crs.to_lonlat(ex, ey)
    return RasterInfo(path=raster_path, crs=str(crs), bounds=[west, south, east, north],
                      bounds_4326=[float(lon.min()), float(lat.min()), float(lon.max()), float(lat.max())],
                      width=w, height=h, bands=b, dtype=_GDAL_TYPE.get(arr.dtype.name, arr.dtype.name))


def _mercator_tags(place: geo.Placement) -> rio.GeoRef:
    return rio.GeoRef({rio.TAG_PIXEL_SCALE: (place.dx, place.dy, 0.0),
                       rio.TAG_TIEPOINT: (0.0, 0.0, 0.0, place.x0, place.y0, 0.0),
                       rio.TAG_GEOKEYS: (1, 1, 0, 3, 1024, 0, 1, 1, 1025, 0, 1, 1, 3072, 0, 1, 3857)})


def reproject_to_web_mercator(input_path: Path, output_path: Path, resample_method: str = "bilinear") -> Path:
    """Writes an RGB GeoTIFF on the EPSG:3857 grid (pixels outside the source are black; the tile
    generator recomputes coverage from the geometry, so no alpha band is stored)."""
    if resample_method != "bilinear":
        raise ValueError("only bilinear resampling is implemented (the reference never passes anything else)")
    input_path, output_path = Path(input_path), Path(output_path)
    arr, _tags, place, crs = _read(input_path)
    rgb = rio._to_u8(arr[..., :3] if arr.shape[2] >= 3 else np.repeat(arr[..., :1], 3, axis=2), 0.0)
    plan = tiles.plan_warp(rgb.shape[1], rgb.shape[0], place, crs)
    eng, lock = _engine_and_lock()
    with lock:                   # the engine's scratch holds a pyramid chain's previous level (see _cut_pyramid)
        out = eng.warp_bilinear_u8(rgb, plan.grid, plan.step, plan.out_h, plan.out_w)
    output_path.parent.mkdir(parents=True, exist_ok=True)
    rio.write_geotiff_rgb(output_path, np.ascontiguousarray(out[..., :3]), _mercator_tags(plan.placement))
    logger.info("Reprojection complete: %s", output_path)
    return output_path


LAST_STATS: dict = {}        # where the last process_raster_to_tiles / generate_xyz_tiles call spent its time (seconds; tools/bench_job.py)


def _cut_pyramid(rgba: np.ndarray, place: geo.Placement, output_dir: Path, min_zoom: int, max_zoom: int, on_device: bool = False) -> None:
    """RGBA raster on the EPSG:3857 grid -> z/x/y.png files (deepest zoom from the raster, the others from their children).
    The levels never leave the device as pixels: each is computed from the previous one's device copy and its PNG files are encoded
    there (s2sr_tiles_write_png: token statistics and bit emission are kernels, the Huffman codes come from the host in between;
    chunk framing, CRC and the file writes run on native host threads).  on_device: `rgba` is what the caller's warp call just left on
    the engine (the caller holds the engine's lock across both): the base level reads that copy.  r04, same 12.8k-tile pyramid: levels fetched and deflated
    by zlib on a Python pool 1.9 s -> native host encoder, one call per 8 tiles 0.6 s -> encoded on the device (this)."""
    import time
    h, w = rgba.shape[:2]
    output_dir.mkdir(parents=True, exist_ok=True)
    eng, lock = _engine_and_lock()
    levels = tiles.plan_levels(place.bounds(w, h), min_zoom, max_zoom)
    prev_lv = None
    t_dev = t_png = 0.0
    # the levels of one pyramid chain through the engine's device copy of the previous level: one pyramid at a time per engine
    with lock:
        for lv in levels:
            t0 = time.perf_counter()
            if prev_lv is None:
                eng.tiles_base_u8(rgba, *tiles.plan_base(lv, place, w, h), fetch=False, on_device=on_device)
            else:
                ox, oy = tiles.overview_offsets(lv, prev_lv)
                eng.tiles_overview_u8((prev_lv.ny, prev_lv.nx), ox, oy, lv.nx, lv.ny, on_device=True, fetch=False)
            t1 = time.perf_counter()
            # <output_dir>/<z>/<x>/<y>.png; tiles outside the raster (alpha 0 everywhere) get no file
            eng.tiles_write_png_xyz(lv.nx, lv.ny, output_dir, lv.zoom, lv.tminx, [geo.xyz_row(lv.tmaxy - j, lv.zoom) for j in range(lv.ny)])
            prev_lv = lv
            t_dev += t1 - t0
            t_png += time.perf_counter() - t1
    LAST_STATS.update(pyramid_level_kernels=t_dev, pyramid_png_on_device_and_files=t_png)


def generate_xyz_tiles(input_path: Path, output_dir: Path, min_zoom: int = 10, max_zoom: int = 16, tile_size: int = 256,
                       resampling: str = "average") -> Path:
    """z/x/y.png (XYZ row order, RGBA) for every tile of zooms min..max that holds data."""
    if tile_size != 256 or resampling != "average":
        raise ValueError("tile_size 256 and average resampling are what the reference uses and what is implemented")
    input_path, output_dir = Path(input_path), Path(output_dir)
    arr, _tags, place, crs = _read(input_path)
    if crs.epsg != 3857:
        raise ValueError(f"{input_path}: {crs}, tiles are cut from an EPSG:3857 raster (reproject_to_web_mercator first)")
    h, w = arr.shape[:2]
    rgba = np.empty((h, w, 4), np.uint8)
    rgba[..., :3] = arr[..., :3] if arr.shape[2] >= 3 else np.repeat(arr[..., :1], 3, axis=2)
    rgba[..., 3] = 255
    _cut_pyramid(rgba, place, output_dir, min_zoom, max_zoom)
    logger.info("Tile generation complete: %s", output_dir)
    return output_dir


def create_tileset_metadata(tiles_dir: Path, bounds_4326: list, min_zoom: int, max_zoom: int,
                            tile_template: str = "/tiles/{z}/{x}/{y}.png") -> dict:
    metadata = {"bounds": bounds_4326, "minzoom": min_zoom, "maxzoom": max_zoom, "tileTemplate": tile_template,
                "attribution": "Sentinel-2 SR via UP42", "format": "png", "tileSize": 256}
    tiles_dir = Path(tiles_dir)
    tiles_dir.mkdir(parents=True, exist_ok=True)
    (tiles_dir / "tileset.json").write_text(json.dumps(metadata, indent=2))
    return metadata


def process_raster_to_tiles(input_path: Path, tiles_dir: Path, min_zoom: int = 10, max_zoom: int = 16) -> dict:
    """Check the CRS, reproject if needed, cut the pyramid, write tileset.json."""
    import time
    input_path, tiles_dir = Path(input_path), Path(tiles_dir)
    LAST_STATS.clear()
    t0 = time.perf_counter()
    arr, _tags, place, crs = _read(input_path)
    h, w, b = arr.shape
    west, south, east, north = place.bounds(w, h)
    t = np.linspace(0.0, 1.0, 21)
    ex = np.concatenate([west + (east - west) * t, np.full(21, east), east - (east - west) * t, np.full(21, west)])
    ey = np.concatenate([np.full(21, north), north - (north - south) * t, np.full(21, south), south + (north - south) * t])
    lon, lat = crs.to_lonlat(ex, ey)
    bounds_4326 = [float(lon.min()), float(lat.min()), float(lon.max()), float(lat.max())]
    rgb = rio._to_u8(arr[..., :3] if b >= 3 else np.repeat(arr[..., :1], 3, axis=2), 0.0)
    t1 = time.perf_counter()
    LAST_STATS["read"] = t1 - t0
    side, err = None, []
    eng, lock = _engine_and_lock()
    with lock:                   # one pyramid chain at a time per engine; held from the warp to the base level that reads its device copy
        if crs.epsg != 3857:
            # the warped raster is written next to the input like the reference does (<stem>_3857.tif, :251-252),
            # but the pyramid is cut from the array in hand, with the coverage mask of the warp as alpha
            plan = tiles.plan_warp(w, h, place, crs)
            t2 = time.perf_counter()
            rgba = eng.warp_bilinear_u8(rgb, plan.grid, plan.step, plan.out_h, plan.out_w)
            t3 = time.perf_counter()
            place = plan.placement
            LAST_STATS.update(warp_plan=t2 - t1, warp_call=t3 - t2)

            def write_3857():            # next to the pyramid, not in front of it: its strips fill the CPUs the device calls leave idle
                try:
                    rio.write_geotiff_rgb(input_path.parent / f"{input_path.stem}_3857.tif", rgba, _mercator_tags(plan.placement))   # alpha dropped per strip
                except BaseException as e:      # noqa: BLE001 -- surfaced below: a failed writer fails the call
                    err.append(e)
            side = threading.Thread(target=write_3857)
            side.start()
        else:
            rgba = np.dstack([rgb, np.full((h, w), 255, np.uint8)])
        t4 = time.perf_counter()
        try:
            _cut_pyramid(np.ascontiguousarray(rgba), place, tiles_dir, min_zoom, max_zoom, on_device=side is not None)
        finally:
            if side is not None:
                t5 = time.perf_counter()
                side.join()
                LAST_STATS["wait_for_3857_tif"] = time.perf_counter() - t5      # what the warped raster's file still needs behind the pyramid
    if err:
        raise err[0]
    LAST_STATS["pyramid"] = time.perf_counter() - t4
    return create_tileset_metadata(tiles_dir, bounds_4326, min_zoom, max_zoom)
