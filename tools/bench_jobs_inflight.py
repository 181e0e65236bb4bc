#!/usr/bin/env python3
"""Jobs per second of one GPU with k /api/wow jobs in flight (k worker threads running process_wow_sr on a 1024x1024 GeoTIFF):
what GpuAdmission(jobs_per_device=k) buys.  A handle serialises its own calls, so k > 1 only overlaps one job's host stages
(read, encoders, file writes) with another's device work.  Usage: tools/bench_jobs_inflight.py [side=1024] [seconds=8]"""
import contextlib
import io
import os
import sys
import tempfile
import threading
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from s2sr import rasterio_lite as rio  # noqa: E402
from s2sr.weights import synthetic_state_dict  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
tmp = Path(tempfile.mkdtemp())
os.environ["S2SR_MODEL_DIR"] = str(tmp / "models")
(tmp / "models").mkdir()
torch.save({"params_ema": {k: torch.from_numpy(v) for k, v in synthetic_state_dict(23, seed=0).items()}}, tmp / "models" / "realesrgan_x4.pth")
yy, xx = np.mgrid[0:side, 0:side]
rng = np.random.default_rng(0)
rgb = np.stack([110 + 70 * np.sin(xx / 23.0 + c) * np.cos(yy / 17.0) + rng.integers(-12, 13, (side, side)) for c in range(3)], -1)
georef = rio.GeoRef({rio.TAG_PIXEL_SCALE: (10.0, 10.0, 0.0), rio.TAG_TIEPOINT: (0.0, 0.0, 0.0, 600000.0, 5100000.0, 0.0),
                     rio.TAG_GEOKEYS: (1, 1, 0, 3, 1024, 0, 1, 1, 1025, 0, 1, 1, 3072, 0, 1, 32633)})
rio.write_geotiff_rgb(tmp / "aoi.tif", np.clip(rgb, 0, 255).astype(np.uint8), georef)
from app.wow_sr import process_wow_sr  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    for _ in range(3):
        process_wow_sr(tmp / "aoi.tif", tmp / "warm")
    rows = []
    for k in (1, 2, 3, 4):
        done = [0] * k
        stop_at = time.time() + seconds

        def worker(t):
            while time.time() < stop_at:
                process_wow_sr(tmp / "aoi.tif", tmp / f"out{t}")
                done[t] += 1
        ts = [threading.Thread(target=worker, args=(t,)) for t in range(k)]
        t0 = time.time()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt = time.time() - t0
        rows.append((k, sum(done), dt))
for k, n, dt in rows:
    print(f"{k} in flight: {n / dt:6.2f} jobs/s ({n} jobs in {dt:.1f} s; {dt / n * 1e3 * k:6.1f} ms per job as seen by its thread)")
