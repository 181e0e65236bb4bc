#!/usr/bin/env python3
"""XYZ pyramid timing on one GPU: a 4096x4096 SR raster (UTM, 2.5 m) -> EPSG:3857 -> z10..18 tiles -> PNG files, the levels
kept on the device and encoded there.  Reports the engine calls and the kernel time from the HIP-event statistics.  The raster
is smooth + noise like an SR output (on pure noise every tile goes to the host encoder: stored blocks)."""
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
import numpy as np  # noqa: E402

from s2sr import geo, native, tiles  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
e = native.Engine(num_block=1)
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:side, 0:side]
rgb = np.clip(np.stack([110 + 70 * np.sin(xx / 93.0 + c) * np.cos(yy / 67.0) for c in range(3)], -1) + rng.integers(-6, 7, (side, side, 3)), 0, 255).astype(np.uint8)
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
import tempfile  # noqa: E402
out_dir = Path(tempfile.mkdtemp())
for warm in (True, False):
    e.reset_kernel_stats()
    t_lv = t_png = 0.0
    k_lv = k_png = 0.0
    ntiles = nfiles = 0
    prev = None
    for lv in levels:
        t0 = time.perf_counter()
        if prev is None:
            e.tiles_base_u8(warped, *tiles.plan_base(lv, plan.placement, plan.out_w, plan.out_h), fetch=False)
        else:
            ox, oy = tiles.overview_offsets(lv, prev)
            e.tiles_overview_u8((prev.ny, prev.nx), ox, oy, lv.nx, lv.ny, on_device=True, fetch=False)
        t1 = time.perf_counter()
        k0 = e.kernel_stats()["misc"]["total_ms"]
        paths = [f"{out_dir}/{int(warm)}/{lv.zoom}/{lv.tminx + i}/{geo.xyz_row(lv.tmaxy - j, lv.zoom)}.png" for j in range(lv.ny) for i in range(lv.nx)]
        wrote = e.tiles_write_png(lv.nx, lv.ny, paths)
        t2 = time.perf_counter()
        k1 = e.kernel_stats()["misc"]["total_ms"]
        t_lv += t1 - t0
        t_png += t2 - t1
        k_png += k1 - k0
        ntiles += lv.nx * lv.ny
        nfiles += int(wrote.sum())
        prev = lv
    k_lv = e.kernel_stats()["misc"]["total_ms"] - k_png
opx = plan.out_h * plan.out_w
print(f"source {side}x{side} -> EPSG:3857 {plan.out_w}x{plan.out_h}; plan {t_plan*1e3:.1f} ms")
print(f"warp: call {t_warp*1e3:.1f} ms (kernel {k_warp:.3f} ms = {opx*(4+12)/k_warp/1e6:.0f} GB/s at 16 B per output px)")
print(f"pyramid z18..10: {ntiles} tiles on the device, level kernels {k_lv:.3f} ms (calls {t_lv*1e3:.1f} ms)")
print(f"PNG files: {nfiles} written in {t_png*1e3:.0f} ms; the two encoder kernels {k_png:.3f} ms = {ntiles*262144*2/k_png/1e6:.0f} GB/s at one read of the tile per kernel")
