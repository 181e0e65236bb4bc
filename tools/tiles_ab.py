#!/usr/bin/env python3
"""A/B of the tile-PNG stage's group size on one box, one process: the z10..18 pyramid of a 4096x4096 SR GeoTIFF, fresh output
directories every time, S2SR_PNG_GROUP_TILES alternating (0 = the whole level as one group, the r04 shape)."""
import os
import shutil
import sys
import tempfile
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
import numpy as np  # noqa: E402

from s2sr import rasterio_lite as rio  # noqa: E402

side = 4096
yy, xx = np.mgrid[0:side, 0:side]
rng = np.random.default_rng(0)
rgb = np.clip(np.stack([110 + 70 * np.sin(xx / 93.0 + c) * np.cos(yy / 67.0) for c in range(3)], -1) + rng.integers(-6, 7, (side, side, 3)), 0, 255).astype(np.uint8)
georef = rio.GeoRef({rio.TAG_PIXEL_SCALE: (2.5, 2.5, 0.0), rio.TAG_TIEPOINT: (0.0, 0.0, 0.0, 600000.0, 5100000.0, 0.0),
                     rio.TAG_GEOKEYS: (1, 1, 0, 3, 1024, 0, 1, 1, 1025, 0, 1, 1, 3072, 0, 1, 32633)})
tmp = Path(tempfile.mkdtemp())
rio.write_geotiff_rgb(tmp / "sr.tif", rgb, georef)
import app.tiling as tiling  # noqa: E402
tiling.process_raster_to_tiles(tmp / "sr.tif", tmp / "warm", 10, 14)
variants = [v for v in sys.argv[1:]] or ["0", "2048", "1024", "4096"]
res = {v: [] for v in variants}
for rep in range(4):
    for v in variants:
        os.environ["S2SR_PNG_GROUP_TILES"] = v
        out = tmp / f"t_{v}_{rep}"
        t0 = time.perf_counter()
        tiling.process_raster_to_tiles(tmp / "sr.tif", out, 10, 18)
        dt = (time.perf_counter() - t0) * 1e3
        res[v].append((dt, tiling.LAST_STATS["pyramid_png_on_device_and_files"] * 1e3))
        shutil.rmtree(out)
for v in variants:
    print(f"group tiles {v:>5s}: pyramid " + " ".join(f"{a:6.1f}" for a, _ in res[v]) + "  | png stage " + " ".join(f"{b:6.1f}" for _, b in res[v]) +
          f"  | median png {sorted(b for _, b in res[v])[len(res[v]) // 2]:.1f} ms")
shutil.rmtree(tmp, ignore_errors=True)
