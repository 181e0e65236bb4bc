#!/usr/bin/env python3
"""PCIe-inclusive rate: s2sr_forward_batch_u8 with HOST buffers in and out (pageable numpy)."""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
import numpy as np
from s2sr import native
from s2sr.synth import synthetic_tiles
from s2sr.weights import synthetic_state_dict
e = native.Engine(num_block=23, precision=native.PREC_F16_HP)
e.load_state_dict(synthetic_state_dict(23, seed=0))
x = synthetic_tiles(32, 256)
e.forward_batch_u8(x)
t0 = time.perf_counter(); n = 3
for _ in range(n):
    y = e.forward_batch_u8(x)
dt = (time.perf_counter() - t0) / n
print(f"host-in/host-out 32 tiles: {dt*1e3:.1f} ms/step -> {32*1.048576/dt:.1f} SR-MP/s (6.3 MB in, 100.7 MB out per step)")
