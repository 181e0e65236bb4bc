#!/usr/bin/env python3
"""Sustained (power-capped) wall time of one conv form with the trace build's DMA ablations:
S2SR_DBG 0 = normal, 8 = no DMA instructions, 2 = slab DMA from one fixed piece, 1 = weights likewise."""
import os
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
os.environ.setdefault("S2SR_LIB", str(REPO / "sentinel2-super-resolution-poc_amd" / "csrc" / "libs2sr_exp.so"))   # stamped builds: make -C csrc EXP=1
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
from s2sr import native  # noqa: E402
from s2sr.weights import synthetic_state_dict  # noqa: E402

e = native.Engine(num_block=6)
e.load_state_dict(synthetic_state_dict(6, seed=0))
N, H, W = 16, 256, 256
for cin, cout in ((160, 32), (64, 32), (192, 64)):
    for dbg in (0, 8, 2, 1, 0):
        os.environ["S2SR_DBG"] = str(dbg)
        for tr in (0, 256):
            if dbg and not tr:
                continue
            r = e.bench_conv(N, H, W, cin, cout, iters=400, trace_wgs=tr)
            us = r[0] if isinstance(r, tuple) else r
            fl = 2.0 * N * H * W * cin * 9 * cout
            print(f"cin={cin} cout={cout} dbg={dbg} trace={tr}: {us:.1f} us  {fl/us/1e6:.0f} TF/s", flush=True)
