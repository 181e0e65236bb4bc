#!/usr/bin/env python3
"""Quick single-GPU timing of the hot path with per-kernel-family HIP-event stats."""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
import torch  # noqa: E402

from s2sr import native  # noqa: E402
from s2sr.weights import synthetic_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--blocks", type=int, default=23)
ap.add_argument("--group", type=int, default=0)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--prof", type=int, default=1)
ap.add_argument("--hp", type=int, default=0, help="precision: 0 fast, 1 hp, 2 fp8")
a = ap.parse_args()

e = native.Engine(num_block=a.blocks, group=a.group, precision=a.hp)
e.load_state_dict(synthetic_state_dict(a.blocks, seed=0))
dev = torch.device("cuda:0")
x = torch.randint(0, 256, (a.batch, a.size, a.size, 3), dtype=torch.uint8, device=dev)
y = torch.empty((a.batch, 4 * a.size, 4 * a.size, 3), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
e.forward_batch_u8_dev(x.data_ptr(), a.batch, a.size, a.size, y.data_ptr(), st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    e.forward_batch_u8_dev(x.data_ptr(), a.batch, a.size, a.size, y.data_ptr(), st)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
mp = a.batch * 16 * a.size * a.size / 1e6
flop = a.batch * a.size * a.size * (35853696 if a.blocks == 23 else 11412864)
print(f"B={a.batch} {a.size}^2 blocks={a.blocks} group={a.group}: {dt*1e3:.2f} ms/step, {mp/dt:.1f} SR-MP/s, "
      f"{a.batch/dt:.1f} tiles/s, {flop/dt/1e12:.1f} TFLOP/s ({flop/dt/2.5e15*100:.1f}% of 2.5 PF)")
if a.prof:
    e.set_profiling(True)
    e.reset_kernel_stats()
    e.forward_batch_u8_dev(x.data_ptr(), a.batch, a.size, a.size, y.data_ptr(), st)
    torch.cuda.synchronize()
    for k, v in e.kernel_stats().items():
        if v["launches"]:
            ms = v["total_ms"]
            print(f"  {k:14s} n={v['launches']:5d} total {ms:9.3f} ms  avg {ms/v['launches']*1e3:9.1f} us  "
                  f"{v['flops']/ms/1e9 if ms else 0:8.1f} TF/s  {v['bytes']/ms/1e6 if ms else 0:8.1f} GB/s(alg)")
