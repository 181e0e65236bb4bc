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
# ... the same bands with a host synchronize behind every band (what s2sr_copy_to_host / d2h_staged do)
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(s):
    for o in range(0, n, band):
        dst[o:o + band].copy_(src[o:o + band], non_blocking=True)
        s.synchronize()
dt = time.perf_counter() - t0
print(f"48-MB bands, synchronize per band: {dt * 1e3:.2f} ms  {n / dt / 1e9:.1f} GB/s")
# ... and through the library: a page-locked array from native.pinned_pool (s2sr_host_alloc: hipHostMallocPortable), s2sr_copy_to_host
from s2sr import native
eng = native.Engine(num_block=1)
out = native.pinned_pool.empty((n,), "uint8")
st = torch.cuda.current_stream().cuda_stream
for label, step in (("one call", n), ("48-MB bands", band)):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for o in range(0, n, step):
            eng.copy_to_host(out[o:o + step], src.data_ptr() + o, st)
        dt = time.perf_counter() - t0
    print(f"s2sr_copy_to_host into a pinned_pool array, {label}: {dt * 1e3:.2f} ms  {n / dt / 1e9:.1f} GB/s")
pag = __import__("numpy").empty(n, "uint8")
t0 = time.perf_counter()
eng.copy_to_host(pag, src.data_ptr(), st)
dt = time.perf_counter() - t0
print(f"s2sr_copy_to_host into a pageable array (staged): {dt * 1e3:.2f} ms  {n / dt / 1e9:.1f} GB/s")
