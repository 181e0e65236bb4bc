#!/usr/bin/env python3
"""Per-wave anatomy of one steady-state stage (diagnostic build, S2SR_DBG=4)."""
import os, sys
from pathlib import Path
import numpy as np
os.environ["S2SR_DBG"] = "4"
REPO = Path(__file__).resolve().parent.parent
os.environ.setdefault("S2SR_LIB", str(REPO / "sentinel2-super-resolution-poc_amd" / "csrc" / "libs2sr_exp.so"))   # stamped builds: make -C csrc EXP=1
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
from s2sr import native
e = native.Engine(num_block=1)
for cin, cout in ((64, 32), (160, 32), (192, 64)):
    us, tr = e.bench_conv(8, 256, 256, cin, cout, iters=5, trace_wgs=256)
    tr = tr.astype(np.int64)
    tr = tr[tr[:, 0] > 0]
    nw = 8 if (tr[:, 4:8] > 0).any() else 4          # 8-wave kernel or the one-wave-per-SIMD trunk kernel
    leave5, leave6, end5 = tr[:, 0:nw], tr[:, 8:8 + nw], tr[:, 16:16 + nw]
    t0 = leave5.min(axis=1, keepdims=True)
    print(f"cin={cin} cout={cout}: stage 5, cycles relative to first wave leaving the barrier (median over {len(tr)} WGs)")
    print("   wave:            " + " ".join(f"{w:6d}" for w in range(nw)))
    print("   leave barrier 5: " + " ".join(f"{int(np.median(leave5[:, w] - t0[:, 0])):6d}" for w in range(nw)))
    print("   last MFMA issued:" + " ".join(f"{int(np.median(end5[:, w] - t0[:, 0])):6d}" for w in range(nw)))
    print("   leave barrier 6: " + " ".join(f"{int(np.median(leave6[:, w] - t0[:, 0])):6d}" for w in range(nw)))
