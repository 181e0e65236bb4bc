#!/usr/bin/env python3
"""Patch-boundary anatomy of the one-wave-per-SIMD trunk kernel (stamped build, S2SR_DBG=16): where the cycles go between
the last stages of one patch and the first stages of the next (epilogue with the matrix pipe idle, ring refill)."""
import os, sys
from pathlib import Path
import numpy as np
os.environ["S2SR_DBG"] = "16"
os.environ["S2SR_TRACE_TIMED"] = "1"
REPO = Path(__file__).resolve().parent.parent
os.environ.setdefault("S2SR_LIB", str(REPO / "sentinel2-super-resolution-poc_amd" / "csrc" / "libs2sr_exp.so"))   # stamped builds: make -C csrc EXP=1
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
from s2sr import native
e = native.Engine(num_block=1, precision=native.PREC_F16_HP)
for cin, cout in ((64, 32), (160, 32), (192, 64)):
    us, tr = e.bench_conv(16, 256, 256, cin, cout, iters=200, trace_wgs=256)
    tr = tr.astype(np.int64)
    tr = tr[(tr[:, 0] > 0) & (tr[:, 16] > 0)]
    w = 0
    bm2, bm1, epi_in, epi_out, b0, b1 = tr[:, 20 + w], tr[:, 4 + w], tr[:, 0 + w], tr[:, 8 + w], tr[:, 16 + w], tr[:, 12 + w]
    med = lambda v: int(np.median(v))
    print(f"cin={cin} cout={cout}: {us:.1f} us/launch, {len(tr)} WGs, wave 0, cycles (median):")
    print(f"   steady stage (barrier to barrier, last two stages of patch 1)   {med(bm1 - bm2):6d}")
    print(f"   last barrier of patch 1 -> epilogue entry                       {med(epi_in - bm1):6d}")
    print(f"   epilogue                                                        {med(epi_out - epi_in):6d}")
    print(f"   epilogue exit -> barrier of patch 2's first stage               {med(b0 - epi_out):6d}")
    print(f"   barrier of first stage -> barrier of second stage               {med(b1 - b0):6d}")
    print(f"   patch boundary total (last barrier of patch 1 -> first of 2)    {med(b0 - bm1):6d}   vs one steady stage {med(bm1 - bm2)}")
