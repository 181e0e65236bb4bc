"""Comparison of a tile pyramid with the GDAL fixture (tests/golden/g10_gdal_tiles.npz, tools/make_gdal_golden.py), shared by the
oracle test (CPU) and the device-pyramid test (GPU)."""
import numpy as np


def compare_with_gdal(golden, ours: dict) -> float:
    """`ours`: {"tile_z_x_y": RGBA array}.  What a map client sees must agree: the same tiles exist, coverage equal up to a
    one-pixel rim, colours close on average.  The resampling definitions differ (this build's footprint average / bilinear warp
    vs GDAL's), so the measured differences are printed and only loosely bounded until real numbers exist."""
    keys = sorted(k for k in golden.files if k.startswith("tile_"))
    assert keys
    assert set(ours) == set(keys), (sorted(set(ours) - set(keys))[:5], sorted(set(keys) - set(ours))[:5])
    worst = 0.0
    for k in keys:
        want, got = golden[k], ours[k]
        a_w, a_g = want[..., 3] > 0, got[..., 3] > 0
        differ = a_w != a_g
        if differ.any():                    # only on the outline: a differing pixel has a pixel of the other kind next to it
            rim = np.zeros_like(differ)
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    rim |= np.roll(np.roll(a_w, dy, 0), dx, 1) != a_w
            assert (differ & ~rim).sum() <= 0.002 * differ.size, k
        both = a_w & a_g
        if both.any():
            worst = max(worst, float(np.abs(want[..., :3][both].astype(int) - got[..., :3][both].astype(int)).mean()))
    print(f"GDAL {golden['gdal_version']}: {len(keys)} tiles, worst mean |difference| over a tile {worst:.2f} grey levels")
    assert worst < 8.0
    return worst
