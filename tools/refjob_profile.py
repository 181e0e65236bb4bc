#!/usr/bin/env python3
"""Where the reference's recorded /api/enhance job (432x576 upload, realesrgan_anime) spends its ~22 ms: cProfile of warm requests."""
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
sys.path.insert(0, str(REPO))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from fastapi.testclient import TestClient  # noqa: E402

from s2sr import rasterio_lite as rio  # noqa: E402
from s2sr.weights import synthetic_state_dict  # noqa: E402

tmp = Path(tempfile.mkdtemp())
os.environ["S2SR_MODEL_DIR"] = str(tmp / "models")
(tmp / "models").mkdir()
for name, nb in (("realesrgan_x4", 23), ("realesrgan_anime", 6)):
    torch.save({"params_ema": {k: torch.from_numpy(v) for k, v in synthetic_state_dict(nb, seed=0).items()}}, tmp / "models" / f"{name}.pth")
rgb = np.ascontiguousarray(np.load(REPO / "tests" / "golden" / "g8_real_image.npz")["img_bgr"][:, :, ::-1])
png = tmp / "u.png"
rio.write_png(png, rgb)
from app.sr_routes import create_app  # noqa: E402
client = TestClient(create_app(tmp / "data", tiler=False, devices=[0]))
b = "BoUnD"
model = sys.argv[1] if len(sys.argv) > 1 else "realesrgan_anime"
body = (f'--{b}\r\nContent-Disposition: form-data; name="model"\r\n\r\n{model}\r\n--{b}\r\nContent-Disposition: form-data; name="image"; filename="u.png"\r\n'
        f'Content-Type: image/png\r\n\r\n').encode() + png.read_bytes() + f"\r\n--{b}--\r\n".encode()


def one():
    r = client.post("/api/enhance", content=body, headers={"content-type": f"multipart/form-data; boundary={b}"})
    assert client.get(f"/api/sr/{r.json()['job_id']}").json()["status"] == "completed"


with contextlib.redirect_stdout(io.StringIO()):
    for _ in range(3):
        one()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        one()
        ts.append((time.perf_counter() - t0) * 1e3)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        one()
    pr.disable()
print(f"{model}: warm requests {['%.1f' % t for t in ts]} ms")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print("\n".join(l[:170] for l in s.getvalue().splitlines()[:80]))
