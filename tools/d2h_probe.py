#!/usr/bin/env python3
"""Device -> page-locked host copy rate of an 805-MB image (a 4096x4096 AOI's output) in one piece, and split over 2 / 4 streams
(one SDMA engine each): what is exposed behind the last window of an enhance_crops job is this copy."""
import sys
import time
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
n = 16384 * 16384 * 3
src = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
dst = torch.empty(n, dtype=torch.uint8, pin_memory=True)
for ways in (1, 2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(ways)]
    cut = [(n * i // ways) & ~4095 for i in range(ways)] + [n]
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                dst[cut[i]:cut[i + 1]].copy_(src[cut[i]:cut[i + 1]], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{ways} stream(s): {dt * 1e3:.2f} ms  {n / dt / 1e9:.1f} GB/s")
# bands of 48 MB one behind the other on one stream (what the finish does)
s = torch.cuda.Stream()
band = 48 << 20
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(s):
    for o in range(0, n, band):
        dst[o:o + band].copy_(src[o:o + band], non_blocking=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"48-MB bands on one stream: {dt * 1e3:.2f} ms  {n / dt / 1e9:.1f} GB/s")
