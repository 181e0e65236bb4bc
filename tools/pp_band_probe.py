#!/usr/bin/env python3
"""Kernel times of the post-process over ONE 16384x16384 image (a 4096x4096 AOI's mosaic): the whole-image launch against the
band-wise route (histograms per 1024-row band, LUTs, apply + sharpen per 48-MB band), by torch events on one stream."""
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
from s2sr import native  # noqa: E402
from s2sr.synth import synthetic_tiles  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
e = native.Engine(num_block=1)
lr = synthetic_tiles(1, side // 4, seed=4321)[0]
img = np.repeat(np.repeat(lr, 4, 0), 4, 1)          # as smooth as an SR output: neighbouring pixels share their L value
x = torch.from_numpy(img).cuda()
y = torch.empty_like(x)
prm = native.pp_wow()
st = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


whole = timed(lambda: e.postprocess_batch_u8_dev(x.data_ptr(), 1, side, side, prm, y.data_ptr(), st))
print(f"whole-image launch: {whole:.3f} ms = {side * side * 9 / whole / 1e6:.0f} GB/s at 9 B/px")
want = y.clone()
for band in (1024, 4096):
    cuts = list(range(0, side, band)) + [side]
    e.pp_band_begin_dev(side, side, prm, 0, st)
    t_h = timed(lambda: [e.pp_band_hist_dev(x.data_ptr(), a, b, st) for a, b in zip(cuts[:-1], cuts[1:])], reps=1)
    t_l = timed(lambda: e.pp_band_lut_dev(st), reps=1)
    t_r = timed(lambda: [e.pp_band_rows_dev(x.data_ptr(), a, b, y.data_ptr(), st) for a, b in zip(cuts[:-1], cuts[1:])], reps=1)
    ok = bool(torch.equal(y, want))
    print(f"bands of {band} rows: hist {t_h:.3f} ms ({len(cuts) - 1} launches), lut {t_l:.3f} ms, apply + sharpen {t_r:.3f} ms; bytes equal the whole-image launch: {ok}")
