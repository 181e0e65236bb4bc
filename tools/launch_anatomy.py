#!/usr/bin/env python3
"""Where the ~8 us of ONE small trunk-conv launch go (single 256x256 tile: one 8x32 patch per workgroup, 256 workgroups).
Stamped build of the single-tile kernel forms (conv_trunk.hip, dbg bit 5): per workgroup entry / prologue DMA issued / stage 0
landed / last stage done / epilogue stores issued / exit, in core cycles (s_memtime) with the constant 100 MHz counter
(s_memrealtime) at entry and exit; next to the launch time hipEvents see around back-to-back launches of the unstamped kernel.
    python tools/launch_anatomy.py [H W]"""
import os
import sys
from pathlib import Path

import numpy as np

os.environ["S2SR_DBG"] = "32"
REPO = Path(__file__).resolve().parent.parent
os.environ.setdefault("S2SR_LIB", str(REPO / "sentinel2-super-resolution-poc_amd" / "csrc" / "libs2sr_exp.so"))   # stamped builds: make -C csrc EXP=1
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
import torch  # noqa: E402,F401
from s2sr import native  # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 256)
e = native.Engine(num_block=1, precision=native.PREC_F16_HP)
print(f"one image of {H}x{W}; times in us; per-workgroup phases are medians over the workgroups that ran")
print(f"{'cin':>4s} {'cout':>4s} {'event us/launch':>15s} {'span':>6s} | {'entry skew':>10s} {'prologue':>8s} {'wait st0':>8s} {'stages':>7s} {'epilogue':>8s} {'drain':>6s} {'in-kernel':>9s} | clock GHz")
for cin, cout in ((64, 32), (96, 32), (128, 32), (160, 32), (192, 64)):
    us, tr = e.bench_conv(1, H, W, cin, cout, iters=300, trace_wgs=256)
    tr = tr[tr[:, 7] > 0].astype(np.int64)
    rt0, rt1 = tr[:, 0], tr[:, 7]
    cyc = tr[:, 1:7]
    ghz = np.median((cyc[:, 5] - cyc[:, 0]) / ((rt1 - rt0) * 10.0))          # cycles per ns
    ph = np.diff(cyc, axis=1) / ghz / 1e3                                    # us per phase and workgroup
    med = np.median(ph, axis=0)
    span = (rt1.max() - rt0.min()) * 0.01
    skew = (np.median(rt0) - rt0.min()) * 0.01
    print(f"{cin:4d} {cout:4d} {us:15.2f} {span:6.2f} | {skew:10.2f} {med[0]:8.2f} {med[1]:8.2f} {med[2]:7.2f} {med[3]:8.2f} {med[4]:6.2f} {ph.sum(axis=1).mean():9.2f} | {ghz:.2f}   ({len(tr)} workgroups)")
e.close()
