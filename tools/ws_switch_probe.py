#!/usr/bin/env python3
"""What a change of image shape costs on one engine (workspace reallocation, dropped graphs): the same two jobs in runs of equal
shapes (A A A A B B B B) and alternating (A B A B ...), 23 blocks, HP mode, device work through enhance_job_u8."""
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
import numpy as np  # noqa: E402

from s2sr import native  # noqa: E402
from s2sr.weights import synthetic_state_dict  # noqa: E402

e = native.Engine(num_block=23, precision=native.PREC_F16_HP)
e.load_state_dict(synthetic_state_dict(23, seed=0))
rng = np.random.default_rng(0)
shapes = [(256, 256), (300, 421), (700, 530), (1024, 1024)]
imgs = {s: rng.integers(0, 256, s + (3,), dtype=np.uint8) for s in shapes}
prm = native.pp_wow()
for s in shapes:                                   # first and second sighting of every shape
    for _ in range(3):
        e.enhance_job_u8(imgs[s], prm)


def run(order):
    t = {}
    for s in order:
        t0 = time.perf_counter()
        e.enhance_job_u8(imgs[s], prm)
        t.setdefault(s, []).append((time.perf_counter() - t0) * 1e3)
    return t


same = run([s for s in shapes for _ in range(6)])
alt = run([s for _ in range(6) for s in shapes])
print(f"{'shape':>12s} {'same shape again (median ms)':>30s} {'after another shape (median ms)':>32s}   allocations so far {e.debug_config()['ws_allocs']}")
for s in shapes:
    a, b = sorted(same[s][1:]), sorted(alt[s][1:])
    print(f"{str(s):>12s} {a[len(a) // 2]:30.1f} {b[len(b) // 2]:32.1f}")
