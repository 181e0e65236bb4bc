#!/usr/bin/env python3
"""Stamp breakdown of the ring kernel's first stages (diagnostic build)."""
import os
import sys
from pathlib import Path
import numpy as np
REPO = Path(__file__).resolve().parent.parent
os.environ.setdefault("S2SR_LIB", str(REPO / "sentinel2-super-resolution-poc_amd" / "csrc" / "libs2sr_exp.so"))   # stamped builds: make -C csrc EXP=1
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
from s2sr import native

e = native.Engine(num_block=1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for cin, cout in ((64, 32), (160, 32), (192, 64)):
    us, tr = e.bench_conv(N, 256, 256, cin, cout, iters=20, trace_wgs=256)
    fl = 2 * 9 * cin * cout * N * 65536
    print(f"cin={cin} cout={cout} N={N}: {us:.1f} us/launch  {fl/us/1e6:.1f} TF/s")
    tot = (tr.astype(np.int64)[:, 21] - tr.astype(np.int64)[:, 20])
    print(f"   TRACE-build kernel time per WG: {np.median(tot[tot>0])/100:.1f} us")
    tr = tr.astype(np.int64)
    tr = tr[tr[:, 0] > 0]
    nst = 9
    med = lambda a, b: float(np.median(tr[:, b] - tr[:, a]))
    print(f"   wgs {len(tr)}  setup+prologue {med(0,1):.0f} first-wait {med(1,2):.0f}", end="")
    for k in range(nst):
        print(f" | s{k}: comp {med(2+2*k, 3+2*k):.0f}", end="")
        if k < nst - 1:
            print(f" wait {med(3+2*k, 4+2*k):.0f}", end="")
    print()
    dt = (tr[:, 23] - tr[:, 22]).astype(float)
    dr = (tr[:, 21] - tr[:, 20]).astype(float)
    print(f"   whole-kernel per WG: {np.median(dt):.0f} shader ticks, {np.median(dr):.0f} realtime ticks (100 MHz) -> clock {np.median(dt/dr)*100:.0f} MHz")
