"""CPU tests of the tile-pyramid host logic: geodesy against published values, tile numbering,
GeoTIFF CRS / placement parsing, and the planner tables (s2sr/tiles.py) against the oracle's
independent pixel-by-pixel geometry (oracle/tiles_ref.py)."""
import numpy as np

from oracle import tiles_ref as ref
from s2sr import geo, tiles


def test_transverse_mercator_published_values_and_roundtrip():
    e, n = geo.tm_forward(0.0, 0.0, 3.0)                    # equator, 3 degrees west of the central meridian
    assert abs(float(e) - 166021.4431) < 1e-3 and abs(float(n)) < 1e-6
    e, n = geo.tm_forward(9.0, 45.0, 9.0)                   # on the central meridian: k0 * meridian arc
    assert abs(float(e) - 500000.0) < 1e-6 and abs(float(n) - 4982950.400) < 1e-3
    e, n = geo.tm_forward(-79.387139, 43.642566, -81.0)     # CN Tower, zone 17N
    assert abs(float(e) - 630084.3) < 0.5 and abs(float(n) - 4833438.5) < 0.5
    rng = np.random.default_rng(0)
    lon, lat = rng.uniform(12.0, 18.0, 2000), rng.uniform(-80.0, 84.0, 2000)
    for south in (False, True):
        E, N = geo.tm_forward(lon, lat, 15.0, south)
        lo2, la2 = geo.tm_inverse(E, N, 15.0, south)
        assert np.abs(lo2 - lon).max() < 1e-11 and np.abs(la2 - lat).max() < 1e-11
    c = geo.CRS(32733)
    x, y = c.from_lonlat(15.2, -33.3)
    assert 0 < float(y) < 1e7 and np.allclose(c.to_lonlat(x, y), (15.2, -33.3), atol=1e-10)


def test_mercator_and_tile_numbering():
    x, y = geo.lonlat_to_mercator(180.0, 0.0)
    assert abs(float(x) - geo.ORIGIN_SHIFT) < 1e-6 and abs(float(y)) < 1e-6
    lon, lat = geo.mercator_to_lonlat(*geo.lonlat_to_mercator(13.4, 52.5))
    assert abs(float(lon) - 13.4) < 1e-12 and abs(float(lat) - 52.5) < 1e-12
    assert abs(geo.resolution(0) - 156543.03392804097) < 1e-6
    tx, ty = geo.meters_to_tile(*(float(v) for v in geo.lonlat_to_mercator(13.4, 52.5)), 10)
    assert (tx, geo.xyz_row(ty, 10)) == (550, 335)           # the well-known z10 tile over Berlin
    w, s, e, n = geo.tile_bounds(tx, ty, 10)
    mx, my = (float(v) for v in geo.lonlat_to_mercator(13.4, 52.5))
    assert w <= mx < e and s <= my < n and abs((e - w) - 256 * geo.resolution(10)) < 1e-6
    assert geo.tile_range((w + 1, s + 1, e - 1, n - 1), 10) == (tx, ty, tx, ty)
    assert geo.tile_range((w + 1, s + 1, e + 1, n + 1), 11) == (2 * tx, 2 * ty, 2 * tx + 2, 2 * ty + 2)


def test_geotiff_crs_and_placement():
    assert geo.epsg_from_geokeys((1, 1, 0, 3, 1024, 0, 1, 1, 1025, 0, 1, 1, 3072, 0, 1, 32633)) == 32633
    assert geo.epsg_from_geokeys((1, 1, 0, 2, 1024, 0, 1, 2, 2048, 0, 1, 4326)) == 4326
    assert geo.epsg_from_geokeys(None) is None
    p = geo.placement_from_tags({33550: (10.0, 10.0, 0.0), 33922: (0.0, 0.0, 0.0, 600000.0, 5100000.0, 0.0)})
    assert p == geo.Placement(600000.0, 5100000.0, 10.0, 10.0)
    assert p.bounds(100, 50) == (600000.0, 5099500.0, 601000.0, 5100000.0)
    q = geo.placement_from_tags({34264: (2.5, 0, 0, 1000.0, 0, -2.5, 0, 2000.0, 0, 0, 0, 0, 0, 0, 0, 1)})
    assert q == geo.Placement(1000.0, 2000.0, 2.5, 2.5)
    assert geo.placement_from_tags({}) is None
    for bad in (2154, 27700):
        try:
            geo.CRS(bad).kind
        except ValueError as e:
            assert "not supported" in str(e)
        else:
            raise AssertionError(bad)


def test_warp_plan_follows_the_projection():
    src = geo.Placement(600000.0, 5100000.0, 2.5, 2.5)
    crs = geo.CRS(32633)
    plan = tiles.plan_warp(400, 300, src, crs)
    st = plan.step
    assert st == 64 and plan.grid.dtype == np.float32
    assert (plan.grid.shape[0] - 1) * st >= plan.out_h - 1 and (plan.grid.shape[1] - 1) * st >= plan.out_w - 1
    # pixel count along the diagonal is preserved (GDAL's suggested-output rule)
    assert abs(np.hypot(plan.out_w, plan.out_h) - np.hypot(400, 300)) < 2.0
    # interpolating the node grid at an off-node pixel agrees with projecting that pixel directly
    oy, ox = 37, 201
    X = plan.placement.x0 + (ox + 0.5) * plan.placement.dx
    Y = plan.placement.y0 - (oy + 0.5) * plan.placement.dy
    lon, lat = geo.mercator_to_lonlat(X, Y)
    sx, sy = crs.from_lonlat(lon, lat)
    u, v = (float(sx) - src.x0) / src.dx - 0.5, (src.y0 - float(sy)) / src.dy - 0.5
    g = plan.grid.astype(np.float64)
    gi, gj, fi, fj = oy // st, ox // st, (oy % st) / st, (ox % st) / st
    top = g[gi, gj] + (g[gi, gj + 1] - g[gi, gj]) * fj
    bot = g[gi + 1, gj] + (g[gi + 1, gj + 1] - g[gi + 1, gj]) * fj
    ui, vi = top + (bot - top) * fi
    assert abs(ui - u) < 0.01 and abs(vi - v) < 0.01
    # ... and on the raster size a job really warps (4096 x 4096 at 2.5 m), anywhere between the nodes, the interpolated source
    # coordinate stays within a thousandth of a pixel of the projected one (the node spacing was chosen for that)
    big = tiles.plan_warp(4096, 4096, src, crs)
    rng = np.random.default_rng(0)
    oy, ox = rng.integers(0, big.out_h, 3000), rng.integers(0, big.out_w, 3000)
    X = big.placement.x0 + (ox + 0.5) * big.placement.dx
    Y = big.placement.y0 - (oy + 0.5) * big.placement.dy
    sx, sy = crs.from_lonlat(*geo.mercator_to_lonlat(X, Y))
    u, v = (sx - src.x0) / src.dx - 0.5, (src.y0 - sy) / src.dy - 0.5
    g = big.grid.astype(np.float64)
    gi, gj, fi, fj = oy // st, ox // st, ((oy % st) / st)[:, None], ((ox % st) / st)[:, None]
    top = g[gi, gj] + (g[gi, gj + 1] - g[gi, gj]) * fj
    bot = g[gi + 1, gj] + (g[gi + 1, gj + 1] - g[gi + 1, gj]) * fj
    uv = top + (bot - top) * fi
    assert np.abs(uv[:, 0] - u).max() < 1e-3 and np.abs(uv[:, 1] - v).max() < 1e-3, (np.abs(uv[:, 0] - u).max(), np.abs(uv[:, 1] - v).max())
    # every corner of the source lands inside the output extent
    w, s, e, n = plan.placement.bounds(plan.out_w, plan.out_h)
    for cx, cy in ((src.x0, src.y0), (src.x0 + 1000.0, src.y0 - 750.0)):
        mx, my = geo.lonlat_to_mercator(*crs.to_lonlat(cx, cy))
        assert w - 1e-6 <= float(mx) <= e + 4.0 and s - 4.0 <= float(my) <= n + 1e-6


def _tiles_from_tables(rgba, col_lo, col_hi, row_lo, row_hi):
    nx, ny = col_lo.size // 256, row_lo.size // 256
    out = np.zeros((ny, nx, 256, 256, 4), np.uint8)
    for gy in range(ny * 256):
        for gx in range(nx * 256):
            if col_hi[gx] < col_lo[gx] or row_hi[gy] < row_lo[gy]:
                continue
            blk = rgba[row_lo[gy]:row_hi[gy] + 1, col_lo[gx]:col_hi[gx] + 1].reshape(1, -1, 4)
            out[gy >> 8, gx >> 8, gy & 255, gx & 255] = ref._mean_valid(blk)[0]
    return out


def test_base_tables_agree_with_the_pixel_geometry():
    """Planner tables (index ranges in float64 pixel coordinates) vs the oracle's metre-space
    membership test, both for down-sampling (footprint of ~3x3 source pixels) and up-sampling."""
    rng = np.random.default_rng(3)
    rgba = rng.integers(0, 256, (90, 120, 4), dtype=np.uint8)
    rgba[..., 3] = np.where(rng.random((90, 120)) < 0.2, 0, 255)
    for dx, zoom in ((3.1, 14), (40.0, 14)):           # z14 = 9.55 m per tile pixel
        place = geo.Placement(1500000.0 + 17.3, 6000000.0 - 5.1, dx, dx)
        lv = tiles.plan_levels(place.bounds(120, 90), zoom, zoom)[0]
        tabs = tiles.plan_base(lv, place, 120, 90)
        got = _tiles_from_tables(rgba, *tabs)
        for j, ty in enumerate(range(lv.tmaxy, lv.tminy - 1, -1)):
            for i, tx in enumerate(range(lv.tminx, lv.tmaxx + 1)):
                want = ref.base_tile(rgba, place.x0, place.y0, place.dx, place.dy, tx, ty, zoom)
                assert np.array_equal(got[j, i], want), (dx, tx, ty)
        assert got[..., 3].any() and not got[..., 3].all()


def test_overview_definition_and_offsets():
    rng = np.random.default_rng(4)
    children = rng.integers(0, 256, (2, 3, 256, 256, 4), dtype=np.uint8)
    children[..., 3] = np.where(rng.random(children.shape[:-1]) < 0.3, 0, 255)
    par = ref.overview(children, ox=-1, oy=0, pnx=2, pny=1)
    # parent (0,0) pixel (200, 130): its 2x2 group lies in child column ox + (2*130 + dx) // 256 = 0, row 1
    grp = np.stack([children[1, 0, (2 * 200 + dy) & 255, (2 * 130 + dx) & 255] for dy in (0, 1) for dx in (0, 1)])
    ok = grp[:, 3] > 0
    if ok.any():
        n = int(ok.sum())
        want = (grp[ok, :3].astype(int).sum(0) + n // 2) // n
        assert np.array_equal(par[0, 0, 200, 130, :3], want) and par[0, 0, 200, 130, 3] == 255
    assert not par[0, 0, :, :128, 3].any()               # west half of the first parent has no child (ox = -1)
    child_lv = tiles.LevelPlan(16, 35733, 42231, 35736, 42232)
    parent_lv = tiles.LevelPlan(15, 17866, 21115, 17868, 21116)
    assert tiles.overview_offsets(parent_lv, child_lv) == (-1, -1)


def test_png_encoder_roundtrips_through_pil():
    import io

    from PIL import Image

    from app.tiling import create_tileset_metadata, encode_png_rgba
    rng = np.random.default_rng(6)
    tile = rng.integers(0, 256, (256, 256, 4), dtype=np.uint8)
    tile[:, 100:, 3] = 0
    for t in (tile, np.zeros((256, 256, 4), np.uint8), np.full((256, 256, 4), 255, np.uint8)):
        im = Image.open(io.BytesIO(encode_png_rgba(t)))
        assert im.mode == "RGBA" and np.array_equal(np.asarray(im), t)


def test_tileset_metadata_contract(tmp_path):
    """tileset.json as the reference writes it (server/app/tiling.py:189-224): keys, template, constants."""
    import json

    from app.tiling import create_tileset_metadata
    m = create_tileset_metadata(tmp_path / "tiles", [16.29, 46.03, 16.31, 46.05], 10, 18)
    assert m == json.loads((tmp_path / "tiles" / "tileset.json").read_text())
    assert m == {"bounds": [16.29, 46.03, 16.31, 46.05], "minzoom": 10, "maxzoom": 18,
                 "tileTemplate": "/tiles/{z}/{x}/{y}.png", "attribution": "Sentinel-2 SR via UP42", "format": "png",
                 "tileSize": 256}


def test_oracle_against_gdal_golden(golden_dir):
    """With tests/golden/g10_gdal_tiles.npz present (tools/make_gdal_golden.py, run where gdalwarp / gdal2tiles.py exist): the
    oracle's pyramid (oracle/tiles_ref.py: bilinear warp, footprint-average base tiles, 2x2-average overviews) for the same
    raster against GDAL's.  Skips without the file (this container has no GDAL; DESIGN.md: "unpinned")."""
    import pytest
    from gdal_compare import compare_with_gdal
    f = golden_dir / "g10_gdal_tiles.npz"
    if not f.exists():
        pytest.skip("tests/golden/g10_gdal_tiles.npz absent: run tools/make_gdal_golden.py where gdalwarp / gdal2tiles.py exist")
    g = np.load(f)
    rgb = g["rgb"]
    h, w = rgb.shape[:2]
    plan = tiles.plan_warp(w, h, geo.Placement(600000.0, 5100000.0, 2.5, 2.5), geo.CRS(32633))
    rgba = ref.warp_bilinear(rgb, plan.grid, plan.step, plan.out_h, plan.out_w)
    place = plan.placement
    levels = tiles.plan_levels(place.bounds(plan.out_w, plan.out_h), int(g["min_zoom"]), int(g["max_zoom"]))
    ours, prev, prev_lv = {}, None, None
    for lv in levels:
        if prev is None:
            cur = np.stack([np.stack([ref.base_tile(rgba, place.x0, place.y0, place.dx, place.dy, lv.tminx + i, lv.tmaxy - j, lv.zoom)
                                      for i in range(lv.nx)]) for j in range(lv.ny)])
        else:
            cur = ref.overview(prev, *tiles.overview_offsets(lv, prev_lv), lv.nx, lv.ny)
        for j in range(lv.ny):
            for i in range(lv.nx):
                if cur[j, i, ..., 3].any():
                    ours[f"tile_{lv.zoom}_{lv.tminx + i}_{geo.xyz_row(lv.tmaxy - j, lv.zoom)}"] = cur[j, i]
        prev, prev_lv = cur, lv
    compare_with_gdal(g, ours)


def test_tiler_takes_the_raster_a_job_just_wrote_from_memory(tmp_path):
    """A job writes <stem>_wow_sr.tif and hands the PATH to the tiler (reference main.py:347-359).  The job's writers ask
    rasterio_lite to remember the array (`remember=True`: they never write to it again); app.tiling._read then gets it from
    memory -- same pixels and geo tags as the file decodes to -- as long as the file's size and mtime are what the write left; a file
    somebody touched, a file written without the request, and a file this process did not write are read from disk."""
    import os
    import time
    import app.tiling as tiling
    from s2sr import rasterio_lite as rio
    from s2sr import tiff_lite
    rio.forget_written()
    rgb = np.random.default_rng(2).integers(0, 256, (120, 160, 3), dtype=np.uint8)
    georef = rio.GeoRef({rio.TAG_PIXEL_SCALE: (2.5, 2.5, 0.0), rio.TAG_TIEPOINT: (0.0, 0.0, 0.0, 600000.0, 5100000.0, 0.0),
                         rio.TAG_GEOKEYS: (1, 1, 0, 3, 1024, 0, 1, 1, 1025, 0, 1, 1, 3072, 0, 1, 32633)})
    p = tmp_path / "job_wow_sr.tif"
    rio.write_outputs(rgb, tmp_path / "job_wow_sr.png", p, georef, remember=True)
    hit = rio.recall_written(p)
    assert hit is not None and hit[0].base is not None and not hit[0].flags.writeable
    disk, dtags = tiff_lite.read_tiff(p)
    assert np.array_equal(hit[0], disk)
    for t in (rio.TAG_PIXEL_SCALE, rio.TAG_TIEPOINT, rio.TAG_GEOKEYS):
        assert tuple(hit[1][t]) == tuple(dtags[t])
    a_mem = tiling._read(p)
    rio.forget_written()
    a_disk = tiling._read(p)
    assert np.array_equal(a_mem[0], a_disk[0]) and a_mem[2] == a_disk[2] and str(a_mem[3]) == str(a_disk[3])
    # not asked to remember -> nothing kept; touched by somebody else -> dropped
    rio.write_geotiff_rgb(tmp_path / "plain.tif", rgb, georef)
    assert rio.recall_written(tmp_path / "plain.tif") is None
    rio.write_geotiff_rgb(p, rgb, georef, remember=True)
    assert rio.recall_written(p) is not None
    time.sleep(0.02)
    os.utime(p, None)
    assert rio.recall_written(p) is None and rio.recall_written(tmp_path / "absent.tif") is None
    # at most two rasters are kept
    for k in range(3):
        rio.write_geotiff_rgb(tmp_path / f"k{k}.tif", rgb, georef, remember=True)
    assert rio.recall_written(tmp_path / "k0.tif") is None and rio.recall_written(tmp_path / "k2.tif") is not None
    rio.forget_written()
