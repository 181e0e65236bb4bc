#!/usr/bin/env python3
"""cProfile of process_wow_sr on the reference's recorded upload (432x576 PNG), anime model, in the calling thread."""
import cProfile
import contextlib
import io
import os
import pstats
import sys
import tempfile
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from s2sr import rasterio_lite as rio  # noqa: E402
from s2sr.weights import synthetic_state_dict  # noqa: E402

tmp = Path(tempfile.mkdtemp())
os.environ["S2SR_MODEL_DIR"] = str(tmp / "models")
(tmp / "models").mkdir()
model = sys.argv[1] if len(sys.argv) > 1 else "realesrgan_anime"
nb = {"realesrgan_x4": 23, "realesrgan_anime": 6}[model]
torch.save({"params_ema": {k: torch.from_numpy(v) for k, v in synthetic_state_dict(nb, seed=0).items()}}, tmp / "models" / f"{model}.pth")
rgb = np.ascontiguousarray(np.load(REPO / "tests" / "golden" / "g8_real_image.npz")["img_bgr"][:, :, ::-1])
png = tmp / "u.png"
rio.write_png(png, rgb)
from app.wow_sr import process_wow_sr  # noqa: E402
with contextlib.redirect_stdout(io.StringIO()):
    for _ in range(3):
        process_wow_sr(png, tmp / "o", enhance_crops=True, model=model)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        process_wow_sr(png, tmp / "o", enhance_crops=True, model=model)
        ts.append((time.perf_counter() - t0) * 1e3)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        process_wow_sr(png, tmp / "o", enhance_crops=True, model=model)
    pr.disable()
print(f"{model}: process_wow_sr warm {['%.1f' % t for t in ts]} ms")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40)
print("\n".join(l[:200] for l in s.getvalue().splitlines()[:70]))
