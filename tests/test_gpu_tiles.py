"""GPU parity of the tile-pyramid kernels (through the C ABI) against the numpy oracle -- bit exact:
integer means, and float32 bilinear steps in a fixed order -- plus the module-level pipeline
(reference surface server/app/tiling.py:227-275) on a synthetic UTM GeoTIFF."""
import json

import numpy as np
import pytest
from PIL import Image

from oracle import tiles_ref as ref
from s2sr import geo, native, tiles
from s2sr import rasterio_lite as rio

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = native.Engine(num_block=1)
    yield e
    e.close()


def _scene(h, w, seed=0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([120 + 90 * np.sin(xx / 11.0 + c) * np.cos(yy / 7.0) + rng.integers(-15, 16, (h, w)) for c in range(3)], -1)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("epsg,origin", [(32633, (600000.0, 5100000.0)), (32733, (300000.0, 6200000.0)), (4326, (13.3, 52.6))])
def test_warp_bit_exact(eng, epsg, origin):
    rgb = _scene(150, 211, seed=epsg)
    px = 2.5 if epsg != 4326 else 2.5e-5
    plan = tiles.plan_warp(211, 150, geo.Placement(origin[0], origin[1], px, px), geo.CRS(epsg))
    got = eng.warp_bilinear_u8(rgb, plan.grid, plan.step, plan.out_h, plan.out_w)
    want = ref.warp_bilinear(rgb, plan.grid, plan.step, plan.out_h, plan.out_w)
    assert got.shape == (plan.out_h, plan.out_w, 4)
    assert np.array_equal(got, want), int((got != want).sum())
    cover = got[..., 3].mean() / 255
    assert 0.9 < cover <= 1.0                                   # the warped outline leaves thin empty wedges at most
    with pytest.raises(native.S2srError):
        eng.warp_bilinear_u8(rgb, plan.grid, 12, plan.out_h, plan.out_w)       # node spacing must be a power of two
    with pytest.raises(native.S2srError):
        eng.warp_bilinear_u8(rgb, plan.grid[:-1], plan.step, plan.out_h, plan.out_w)   # grid too small for the output


def test_base_and_overview_bit_exact(eng):
    rng = np.random.default_rng(9)
    rgba = np.concatenate([_scene(300, 420, seed=2), np.full((300, 420, 1), 255, np.uint8)], -1)
    rgba[..., 3] = np.where(rng.random((300, 420)) < 0.15, 0, 255)
    place = geo.Placement(1500017.3, 5999994.9, 3.1, 3.1)
    levels = tiles.plan_levels(place.bounds(420, 300), 12, 15)
    base = eng.tiles_base_u8(rgba, *tiles.plan_base(levels[0], place, 420, 300))
    lv = levels[0]
    assert base.shape == (lv.ny, lv.nx, 256, 256, 4)
    for j, i in ((0, 0), (lv.ny - 1, lv.nx - 1)):
        want = ref.base_tile(rgba, place.x0, place.y0, place.dx, place.dy, lv.tminx + i, lv.tmaxy - j, lv.zoom)
        assert np.array_equal(base[j, i], want)
    cur, cur_lv = base, lv
    for nxt in levels[1:]:
        ox, oy = tiles.overview_offsets(nxt, cur_lv)
        got = eng.tiles_overview_u8(cur, ox, oy, nxt.nx, nxt.ny, on_device=(nxt.zoom % 2 == 0))   # from the host copy / the device copy
        assert np.array_equal(got, ref.overview(cur, ox, oy, nxt.nx, nxt.ny)), nxt.zoom
        cur, cur_lv = got, nxt
    assert cur[..., 3].any()
    with pytest.raises(native.S2srError, match="did not leave a tile level"):      # a level of another size is not "the previous level"
        eng.tiles_overview_u8(np.zeros((cur.shape[0] + 1, cur.shape[1], 256, 256, 4), np.uint8), 0, 0, 1, 1, on_device=True)
    eng.postprocess_u8(np.zeros((64, 64, 3), np.uint8), native.pp_wow())           # any other call takes the scratch
    with pytest.raises(native.S2srError, match="did not leave a tile level"):
        eng.tiles_overview_u8(cur, 0, 0, 1, 1, on_device=True)
    bad = tiles.plan_base(lv, place, 420, 300)
    bad[1][5] = 420                                             # a footprint that leaves the raster must be refused
    with pytest.raises(native.S2srError):
        eng.tiles_base_u8(rgba, *bad)


def test_level_pngs_encoded_on_the_device(eng, tmp_path):
    """s2sr_tiles_write_png: the level stays on the device, its PNG files must decode to exactly the tiles a fetch returns --
    smooth tiles (Huffman blocks with runs), noise tiles (handed to the host encoder: stored blocks), half-covered tiles,
    fully transparent ones (no file), skipped paths."""
    from PIL import Image
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:700, 0:900]
    rgba = np.empty((700, 900, 4), np.uint8)
    rgba[..., :3] = np.clip(120 + 80 * np.sin(xx / 31.0)[..., None] * np.cos(yy / 17.0)[..., None] + rng.integers(-3, 4, (700, 900, 3)), 0, 255)
    rgba[:, 600:, :3] = rng.integers(0, 256, (700, 300, 3))                     # a noisy part
    rgba[..., 3] = 255
    rgba[:200, :300, 3] = 0                                                     # a hole
    place = geo.Placement(1500017.3, 5999994.9, 0.6, 0.6)
    levels = tiles.plan_levels(place.bounds(900, 700), 15, 18)
    lv = levels[0]
    want = eng.tiles_base_u8(rgba, *tiles.plan_base(lv, place, 900, 700))
    prev = None
    for k, cur_lv in enumerate(levels):
        if k == 0:
            assert eng.tiles_base_u8(rgba, *tiles.plan_base(lv, place, 900, 700), fetch=False) is None
        else:
            ox, oy = tiles.overview_offsets(cur_lv, prev)
            want = ref.overview(want, ox, oy, cur_lv.nx, cur_lv.ny)
            assert eng.tiles_overview_u8((prev.ny, prev.nx), ox, oy, cur_lv.nx, cur_lv.ny, on_device=True, fetch=False) is None
        paths = [tmp_path / str(cur_lv.zoom) / str(i) / f"{j}.png" for j in range(cur_lv.ny) for i in range(cur_lv.nx)]
        if k == 0:
            paths[1] = None
        wrote = eng.tiles_write_png(cur_lv.nx, cur_lv.ny, paths)
        n_files = 0
        for j in range(cur_lv.ny):
            for i in range(cur_lv.nx):
                p = paths[j * cur_lv.nx + i]
                has = bool(want[j, i, ..., 3].any()) and p is not None
                assert bool(wrote[j, i]) == has and (p is None or p.exists() == has), (cur_lv.zoom, j, i)
                if has:
                    im = Image.open(p)
                    im.verify()
                    assert np.array_equal(np.asarray(Image.open(p)), want[j, i]), (cur_lv.zoom, j, i)
                    n_files += 1
        assert n_files > 0
        if k == 0:      # Huffman streams from the device: a smooth tile is a fraction of its 256 KB
            assert min(p.stat().st_size for p in paths if p is not None and p.exists()) < 150_000
        prev = cur_lv
    assert not want[..., 3].all() or True
    # A level made of the raster's own pixels (footprint tables of one pixel each): noise, a flat and a transparent tile.  Level
    # alpha is 0 or 255, so even noise keeps a compressible channel and stays on the device route; the host-encoder route (taken
    # when stored blocks would be smaller, or a block header outgrows its slot) is forced once through its flag.
    noise = rng.integers(0, 256, (512, 512, 4), dtype=np.uint8)
    noise[..., 3] = 255
    noise[256:, :256, :3] = 77
    noise[256:, 256:, 3] = 0
    ident = np.arange(512, dtype=np.int32)
    got = eng.tiles_base_u8(noise, ident, ident, ident, ident)
    assert np.array_equal(got[1, 0], noise[256:, :256]) and np.array_equal(got[0, 1], noise[:256, 256:])
    for force_host in (False, True):
        eng.tiles_base_u8(noise, ident, ident, ident, ident, fetch=False)
        npaths = [tmp_path / f"noise{int(force_host)}" / f"{j}_{i}.png" for j in range(2) for i in range(2)]
        assert eng.tiles_write_png(2, 2, npaths, host_encoder=force_host).tolist() == [[1, 1], [1, 0]]
        for q, (j, i) in zip(npaths[:3], ((0, 0), (0, 1), (1, 0))):
            assert np.array_equal(np.asarray(Image.open(q)), got[j, i]), (force_host, j, i)
        assert npaths[2].stat().st_size < 2_000 and not npaths[3].exists()
    eng.postprocess_u8(np.zeros((64, 64, 3), np.uint8), native.pp_wow())           # any other call takes the scratch
    with pytest.raises(native.S2srError, match="did not leave a tile level"):
        eng.tiles_write_png(1, 1, [tmp_path / "x.png"])


def test_tile_png_kernel_forms_write_the_same_bytes(eng, tmp_path):
    """The two kernels of s2sr_tiles_write_png exist in two forms -- one thread walks one row (the first, kept as the check), one
    wave per row with the run structure from a ballot and two shuffles (the default) -- and must produce the same token stream:
    the files are compared byte for byte.  Rows are built from their Sub-filtered bytes: zero stretches of every length around
    the interesting ones (1, 2, 3 bytes: literals or a match; 258 + 1 / 2 / 3: the chunk rule; beyond 516 and 774; across the
    16-byte lane boundaries), noise, flat rows, runs that start at the filter byte."""
    from PIL import Image
    rng = np.random.default_rng(17)
    H = W = 512
    lens = [1, 2, 3, 4, 5, 14, 15, 16, 17, 18, 31, 32, 33, 63, 64, 65, 255, 256, 257, 258, 259, 260, 261, 262, 300, 514, 515, 516, 517, 518, 519,
            520, 773, 774, 775, 776, 777, 1000, 1020]
    img = np.zeros((H, W, 4), np.uint8)
    for y in range(H):
        g = np.zeros((W, 3), np.int64)                       # per-pixel colour deltas: the filtered bytes (alpha's delta is 0)
        kind = y % 5
        if kind == 0:                                        # zero stretches of chosen lengths between single non-zero bytes
            flat = np.zeros(W * 3, np.int64)
            pos = int(rng.integers(0, 8))
            while pos < flat.size:
                flat[pos] = int(rng.integers(1, 256))
                pos += 1 + int(lens[int(rng.integers(len(lens)))]) * 3 // 4
            g = flat.reshape(W, 3)
        elif kind == 1:
            g = rng.integers(0, 256, (W, 3))                 # noise
        elif kind == 2:
            g[rng.random(W) < 0.03] = rng.integers(1, 256, 3)     # sparse steps: long flat stretches
        elif kind == 3:
            g[:] = 1 if y % 2 else 0                         # all-ones deltas (a run that starts at the filter byte 1) / all zero
            g[0] = (1, 1, 1)
        else:
            g[::int(rng.integers(2, 70))] = rng.integers(0, 3, 3)
        img[y, :, :3] = np.cumsum(g, axis=0) % 256
    img[..., 3] = 255
    ident = np.arange(512, dtype=np.int32)
    got = eng.tiles_base_u8(img, ident, ident, ident, ident)
    assert np.array_equal(got[0, 0], img[:256, :256])
    files = {}
    for form in (False, True):
        eng.tiles_base_u8(img, ident, ident, ident, ident, fetch=False)
        paths = [tmp_path / f"form{int(form)}" / f"{j}_{i}.png" for j in range(2) for i in range(2)]
        assert eng.tiles_write_png(2, 2, paths, row_threads=form).all()
        files[form] = [q.read_bytes() for q in paths]
        for q, (j, i) in zip(paths, ((0, 0), (0, 1), (1, 0), (1, 1))):
            assert np.array_equal(np.asarray(Image.open(q)), got[j, i]), (form, j, i)
    assert files[False] == files[True]


def test_tile_png_groups_pipeline_writes_the_same_files(eng, tmp_path):
    """A level goes through s2sr_tiles_write_png in groups (~2048 tiles; a z18 level is five): all statistics kernels queued up front,
    then per group Huffman codes on the host, upload + emit kernel, and -- while that runs -- the previous group's streams back and
    its files written; streams ping-pong between two device buffers, statistics and plans live at per-group offsets of one
    page-locked block.  With S2SR_PNG_SMALL_GROUPS the groups are 3 tiles: a 5 x 4 level (smooth, noisy, half-covered, transparent
    and skipped tiles; ragged last group) must give the files of the one-group form, byte for byte, and the same `written`."""
    from PIL import Image
    rng = np.random.default_rng(23)
    H, W = 4 * 256, 5 * 256
    yy, xx = np.mgrid[0:H, 0:W]
    rgba = np.empty((H, W, 4), np.uint8)
    rgba[..., :3] = np.clip(120 + 80 * np.sin(xx / 37.0)[..., None] * np.cos(yy / 23.0)[..., None] + rng.integers(-4, 5, (H, W, 3)), 0, 255)
    rgba[256:512, 512:768, :3] = rng.integers(0, 256, (256, 256, 3))           # a noise tile
    rgba[..., 3] = 255
    rgba[768:, :256, 3] = 0                                                     # a transparent tile: no file
    rgba[:128, 1024:, 3] = 0                                                    # a half-covered one
    iy, ix = np.arange(H, dtype=np.int32), np.arange(W, dtype=np.int32)
    want = eng.tiles_base_u8(rgba, ix, ix, iy, iy)
    assert want.shape == (4, 5, 256, 256, 4) and np.array_equal(want[1, 2], rgba[256:512, 512:768])
    out = {}
    for small in (False, True):
        eng.tiles_base_u8(rgba, ix, ix, iy, iy, fetch=False)
        paths = [tmp_path / f"g{int(small)}" / f"{j}_{i}.png" for j in range(4) for i in range(5)]
        paths[7] = None
        wrote = eng.tiles_write_png(5, 4, paths, small_groups=small)
        out[small] = (wrote.copy(), [None if q is None or not q.exists() else q.read_bytes() for q in paths])
        for k, q in enumerate(paths):
            j, i = divmod(k, 5)
            has = q is not None and bool(want[j, i, ..., 3].any())
            assert bool(wrote[j, i]) == has and (q is None or q.exists() == has), (small, j, i)
            if has:
                assert np.array_equal(np.asarray(Image.open(q)), want[j, i]), (small, j, i)
    assert np.array_equal(out[False][0], out[True][0]) and out[False][1] == out[True][1]
    assert not out[True][0][3, 0]            # the transparent tile


def test_process_raster_to_tiles_end_to_end(tmp_path):
    import app.tiling as tiling
    rgb = _scene(240, 320, seed=5)
    georef = rio.GeoRef({rio.TAG_PIXEL_SCALE: (2.5, 2.5, 0.0), rio.TAG_TIEPOINT: (0.0, 0.0, 0.0, 600000.0, 5100000.0, 0.0),
                         rio.TAG_GEOKEYS: (1, 1, 0, 3, 1024, 0, 1, 1, 1025, 0, 1, 1, 3072, 0, 1, 32633)})
    src = tmp_path / "aoi_wow_sr.tif"
    rio.write_geotiff_rgb(src, rgb, georef)
    info = tiling.get_raster_info(src)
    assert (info.crs, info.width, info.height, info.bands, info.dtype) == ("EPSG:32633", 320, 240, 3, "Byte")
    cx, cy = np.array([600000.0, 600800.0, 600000.0, 600800.0]), np.array([5100000.0, 5100000.0, 5099400.0, 5099400.0])
    lon, lat = geo.CRS(32633).to_lonlat(cx, cy)                 # grid convergence: the extremes sit at different corners
    assert np.allclose(info.bounds_4326, [lon.min(), lat.min(), lon.max(), lat.max()], atol=1e-6)
    meta = tiling.process_raster_to_tiles(src, tmp_path / "tiles", min_zoom=12, max_zoom=16)
    assert meta == json.loads((tmp_path / "tiles" / "tileset.json").read_text())
    assert meta["minzoom"] == 12 and meta["maxzoom"] == 16 and meta["tileSize"] == 256 and meta["format"] == "png"
    assert (tmp_path / "aoi_wow_sr_3857.tif").exists()
    for z in range(12, 17):
        pngs = list((tmp_path / "tiles" / str(z)).glob("*/*.png"))
        assert pngs, z
        t = np.asarray(Image.open(pngs[0]))
        assert t.shape == (256, 256, 4) and t[..., 3].any()
    # the tile that holds the raster's north-west corner at z16, pixel by pixel against the oracle
    w3857 = tiling.get_raster_info(tmp_path / "aoi_wow_sr_3857.tif")
    assert w3857.crs == "EPSG:3857"
    arr, georef3857 = rio.read_rgb_u8(tmp_path / "aoi_wow_sr_3857.tif")
    place = geo.placement_from_tags(georef3857.tags)
    plan = tiles.plan_warp(320, 240, geo.Placement(600000.0, 5100000.0, 2.5, 2.5), geo.CRS(32633))
    alpha = ref.warp_bilinear(rgb, plan.grid, plan.step, plan.out_h, plan.out_w)[..., 3]
    rgba = np.dstack([arr, alpha])
    tx, ty = geo.meters_to_tile(place.x0 + 200.0, place.y0 - 200.0, 16)
    want = ref.base_tile(rgba, place.x0, place.y0, place.dx, place.dy, tx, ty, 16)
    got = np.asarray(Image.open(tmp_path / "tiles" / "16" / str(tx) / f"{geo.xyz_row(ty, 16)}.png"))
    assert np.array_equal(got, want)
    # a raster that is already EPSG:3857 is tiled as is; anything else must be reprojected first
    with pytest.raises(ValueError, match="EPSG:32633"):
        tiling.generate_xyz_tiles(src, tmp_path / "nope")
    tiling.generate_xyz_tiles(tmp_path / "aoi_wow_sr_3857.tif", tmp_path / "tiles2", min_zoom=15, max_zoom=15)
    assert list((tmp_path / "tiles2" / "15").glob("*/*.png"))


def test_http_wow_job_tiles_by_default(monkeypatch, tmp_path):
    """POST /api/wow with the harness' default tiler: SR GeoTIFF (UTM, pixel size / 4) -> EPSG:3857 ->
    z10..18 pyramid under data/tiles_wow with tileset.json, as run_wow_job does (main.py:347-359)."""
    import torch
    from fastapi.testclient import TestClient

    from app.sr_routes import create_app
    from s2sr.weights import synthetic_state_dict
    monkeypatch.setenv("S2SR_MODEL_DIR", str(tmp_path / "models"))
    (tmp_path / "models").mkdir()
    sd = {k: torch.from_numpy(v) for k, v in synthetic_state_dict(23, seed=0).items()}
    torch.save({"params_ema": sd}, tmp_path / "models" / "realesrgan_x4.pth")
    rgb = _scene(24, 32, seed=8)
    georef = rio.GeoRef({rio.TAG_PIXEL_SCALE: (10.0, 10.0, 0.0), rio.TAG_TIEPOINT: (0.0, 0.0, 0.0, 600000.0, 5100000.0, 0.0),
                         rio.TAG_GEOKEYS: (1, 1, 0, 3, 1024, 0, 1, 1, 1025, 0, 1, 1, 3072, 0, 1, 32633)})
    (tmp_path / "data" / "source").mkdir(parents=True)
    rio.write_geotiff_rgb(tmp_path / "data" / "source" / "s2.tif", rgb, georef)
    c = TestClient(create_app(tmp_path / "data"))
    r = c.post("/api/wow", json={"auto_fetch": False})
    st = c.get(f"/api/sr/{r.json()['job_id']}").json()
    assert st["status"] == "completed", st
    tdir = tmp_path / "data" / "tiles_wow"
    assert st["result"]["tiles_dir"] == str(tdir)
    meta = json.loads((tdir / "tileset.json").read_text())
    assert (meta["minzoom"], meta["maxzoom"]) == (10, 18)
    for z in (10, 14, 18):
        pngs = list((tdir / str(z)).glob("*/*.png"))
        assert pngs and np.asarray(Image.open(pngs[0])).shape == (256, 256, 4)
    lon, lat = geo.CRS(32633).to_lonlat(600160.0, 5099880.0)       # centre of the 320 m x 240 m scene
    assert meta["bounds"][0] < float(lon) < meta["bounds"][2] and meta["bounds"][1] < float(lat) < meta["bounds"][3]


def test_gpu_pyramid_against_gdal_golden(golden_dir, tmp_path):
    """With tests/golden/g10_gdal_tiles.npz present (tools/make_gdal_golden.py, run where GDAL's tools are installed): the pyramid
    this build writes for the same GeoTIFF against gdalwarp + gdal2tiles.py's.  Skips without the file (this container has no GDAL)."""
    import app.tiling as tiling
    f = golden_dir / "g10_gdal_tiles.npz"
    if not f.exists():
        pytest.skip("tests/golden/g10_gdal_tiles.npz absent: run tools/make_gdal_golden.py where gdalwarp / gdal2tiles.py exist")
    g = np.load(f)
    georef = rio.GeoRef({rio.TAG_PIXEL_SCALE: (2.5, 2.5, 0.0), rio.TAG_TIEPOINT: (0.0, 0.0, 0.0, 600000.0, 5100000.0, 0.0),
                         rio.TAG_GEOKEYS: (1, 1, 0, 3, 1024, 0, 1, 1, 1025, 0, 1, 1, 3072, 0, 1, 32633)})
    rio.write_geotiff_rgb(tmp_path / "aoi.tif", g["rgb"], georef)
    tiling.process_raster_to_tiles(tmp_path / "aoi.tif", tmp_path / "tiles", int(g["min_zoom"]), int(g["max_zoom"]))
    from gdal_compare import compare_with_gdal
    ours = {f"tile_{q.parts[-3]}_{q.parts[-2]}_{q.stem}": np.asarray(Image.open(q).convert("RGBA")) for q in (tmp_path / "tiles").glob("*/*/*.png")}
    compare_with_gdal(g, ours)
