#!/usr/bin/env python3
"""Is the fp16 trunk HBM-bound?  One RDB-shaped conv launched back to back on the same buffers for N images: for small N
the operands stay in the 256 MiB Infinity Cache across launches, for large N they stream from HBM."""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
import torch  # noqa: F401,E402
from s2sr import native  # noqa: E402
e = native.Engine(num_block=1)
for cin, cout in ((160, 32), (64, 32), (192, 64)):
    for N in (4, 8, 12, 16, 24, 32, 48):
        fl = 2.0 * N * 256 * 256 * cin * 9 * cout
        iters = max(50, int(1.0e6 / (fl / 0.9e9)))
        us, _ = e.bench_conv(N, 256, 256, cin, cout, iters=iters)
        mb = N * 258 * 258 * 32 * (cin // 16 + cout // 16) / 1e6
        print(f"cin={cin} cout={cout} N={N:3d}: {us:8.1f} us  {us / N:6.2f} us/img  {fl / us / 1e6:7.0f} TF/s  operands {mb:6.0f} MB", flush=True)
