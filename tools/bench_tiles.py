#!/usr/bin/env python3
"""XYZ pyramid timing on one GPU: a 4096x4096 SR raster (UTM, 2.5 m) -> EPSG:3857 -> z10..18 tile arrays.
Reports the engine calls (host buffers in/out, kernel time from the HIP-event statistics) and, separately,
what PNG encoding of the tiles costs on the host."""
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
import numpy as np  # noqa: E402
from app.tiling import encode_png_rgba  # noqa: E402

from s2sr import geo, native, tiles  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
e = native.Engine(num_block=1)
rng = np.random.default_rng(0)
rgb = rng.integers(0, 256, (side, side, 3), dtype=np.uint8)
src = geo.Placement(600000.0, 5100000.0, 2.5, 2.5)
t0 = time.perf_counter()
plan = tiles.plan_warp(side, side, src, geo.CRS(32633))
t_plan = time.perf_counter() - t0
e.warp_bilinear_u8(rgb[:64, :64], plan.grid[:8, :8], plan.step, 32, 32)      # warm-up
e.set_profiling(1)
e.reset_kernel_stats()
t0 = time.perf_counter()
warped = e.warp_bilinear_u8(rgb, plan.grid, plan.step, plan.out_h, plan.out_w)
t_warp = time.perf_counter() - t0
k_warp = e.kernel_stats()["misc"]["total_ms"]
levels = tiles.plan_levels(plan.placement.bounds(plan.out_w, plan.out_h), 10, 18)
e.reset_kernel_stats()
t0 = time.perf_counter()
arrays = []
cur, cur_lv = None, None
for lv in levels:
    if cur is None:
        cur = e.tiles_base_u8(warped, *tiles.plan_base(lv, plan.placement, plan.out_w, plan.out_h))
    else:
        ox, oy = tiles.overview_offsets(lv, cur_lv)
        cur = e.tiles_overview_u8(cur, ox, oy, lv.nx, lv.ny)
    cur_lv = lv
    arrays.append(cur)
t_tiles = time.perf_counter() - t0
k_tiles = e.kernel_stats()["misc"]["total_ms"]
ntiles = sum(a.shape[0] * a.shape[1] for a in arrays)
from concurrent.futures import ThreadPoolExecutor  # noqa: E402
import os  # noqa: E402
todo = [t for a in arrays for t in a.reshape(-1, 256, 256, 4) if t[..., 3].any()]
nenc = len(todo)
t0 = time.perf_counter()
nthr = min(16, os.cpu_count() or 4)
with ThreadPoolExecutor(max_workers=nthr) as pool:
    total_bytes = sum(len(b) for b in pool.map(encode_png_rgba, todo, chunksize=8))
t_png = time.perf_counter() - t0
opx = plan.out_h * plan.out_w
print(f"source {side}x{side} -> EPSG:3857 {plan.out_w}x{plan.out_h}; plan {t_plan*1e3:.1f} ms")
print(f"warp: call {t_warp*1e3:.1f} ms (kernel {k_warp:.3f} ms = {opx*(4+12)/k_warp/1e6:.0f} GB/s at 16 B per output px)")
print(f"pyramid z18..10: {ntiles} tiles, calls {t_tiles*1e3:.1f} ms (kernels {k_tiles:.3f} ms)")
print(f"PNG encode on the host (zlib level 1 + Z_RLE, {nthr} threads): {nenc} tiles, {total_bytes/1e6:.0f} MB in {t_png*1e3:.0f} ms")
