#!/usr/bin/env python3
"""In-kernel stamp breakdown of one RDB conv (diagnostic build; read shares, not lengths)."""
import sys
from pathlib import Path
import numpy as np
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
from s2sr import native

e = native.Engine(num_block=1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for cin, cout in ((64, 32), (160, 32), (192, 64)):
    us, tr = e.bench_conv(N, 256, 256, cin, cout, iters=20, trace_wgs=4096)
    fl = 2 * 9 * cin * cout * N * 65536
    print(f"cin={cin} cout={cout} N={N}: {us:.1f} us/launch  {fl/us/1e6:.1f} TF/s")
    tr = tr.astype(np.int64)
    ok = tr[:, 0] > 0
    tr = tr[ok]
    t0 = tr[:, 0].min()
    nch = cin // 32
    d = lambda a, b: (tr[:, b] - tr[:, a])
    print(f"   wgs traced {len(tr)}; kernel span {(tr[:,23].max()-t0)/100:.1f} us (s_memtime 100MHz ticks)")
    print(f"   setup {np.median(d(0,1)):.0f}  first-chunk wait {np.median(d(1,2)):.0f}", end="")
    for c in range(nch):
        print(f" | c{c} compute {np.median(d(2+2*c, 3+2*c)):.0f} wait {np.median(d(3+2*c, 4+2*c)):.0f}", end="")
    print(f" | epilogue {np.median(d(2+2*nch, 23)):.0f}   total {np.median(d(0,23)):.0f} ticks")
    # distribution of wg start times (waves of workgroups)
    st = np.sort(tr[:, 0] - t0)
    print("   wg start ticks pctl 0/25/50/75/100:", [int(np.percentile(st, q)) for q in (0, 25, 50, 75, 100)])
