#!/usr/bin/env python3
"""Where reading a 4096x4096 LZW GeoTIFF (a job's sr_tif) spends its time on this host."""
import sys
import tempfile
import time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
import numpy as np  # noqa: E402
from s2sr import hostpool, native, tiff_lite  # noqa: E402
from s2sr import rasterio_lite as rio  # noqa: E402
side = 4096
yy, xx = np.mgrid[0:side, 0:side]
rng = np.random.default_rng(0)
rgb = np.clip(np.stack([110 + 70 * np.sin(xx / 93.0 + c) * np.cos(yy / 67.0) for c in range(3)], -1) + rng.integers(-2, 3, (side, side, 3)), 0, 255).astype(np.uint8)
geo = rio.GeoRef({rio.TAG_PIXEL_SCALE: (2.5, 2.5, 0.0), rio.TAG_TIEPOINT: (0.0, 0.0, 0.0, 6e5, 5.1e6, 0.0)})
d = Path(tempfile.mkdtemp())
rio.write_geotiff_rgb(d / "a.tif", rgb, geo)
print(f"file {(d / 'a.tif').stat().st_size / 1e6:.0f} MB, pool {hostpool.workers()} workers")
for rep in range(3):
    t0 = time.perf_counter()
    raw = (d / "a.tif").read_bytes()
    t1 = time.perf_counter()
    out = np.empty((side, side, 3), np.uint8)
    t2 = time.perf_counter()
    arr, tags = tiff_lite.read_tiff(d / "a.tif")
    t3 = time.perf_counter()
    # decode only: strips straight into a pre-touched array
    offs, cnts = tags[273], tags[279]
    ru8 = np.frombuffer(raw, np.uint8)
    out[:] = 0
    t4 = time.perf_counter()
    list(hostpool.pool().map(lambda k: native.tiff_lzw_decode_into(ru8, int(offs[k]), int(cnts[k]), out[64 * k:64 * k + 64]), range(len(offs))))
    t5 = time.perf_counter()
    print(f"rep {rep}: read_bytes {1e3 * (t1 - t0):.1f} ms, np.empty {1e3 * (t2 - t1):.2f}, read_tiff {1e3 * (t3 - t2):.1f}, decode of {len(offs)} strips into a touched array {1e3 * (t5 - t4):.1f}; equal {np.array_equal(arr, rgb)}")
import shutil  # noqa: E402
shutil.rmtree(d, ignore_errors=True)
