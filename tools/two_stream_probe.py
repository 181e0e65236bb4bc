#!/usr/bin/env python3
"""Probe: E engines on E streams side by side (each its own workspace, graphs and HIP stream), B tiles each per step, launch groups
of G -- do small, Infinity-Cache-sized groups pay once another queue's workgroups fill the CUs a draining launch leaves?
(profiles/r05_mfma_ceiling.txt: the same LDS-DMA bytes from cache sustain 1.32 PFLOP/s against 1.0 from HBM; a launch group of 4
images has a dense tensor of 100 MB, of 16 images 403 MB against 256 MB of Infinity Cache.)

    python3 tools/two_stream_probe.py --engines 2 --batch 16 --group 4 --steps 10
"""
import argparse
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
sys.path.insert(0, str(REPO))
import torch  # noqa: E402

from bench import ClockSampler  # noqa: E402
from s2sr import native  # noqa: E402
from s2sr.weights import synthetic_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--engines", type=int, default=2)
ap.add_argument("--batch", type=int, default=16, help="tiles per engine and step")
ap.add_argument("--group", type=int, default=4)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=4)
ap.add_argument("--hp", type=int, default=1)
a = ap.parse_args()

dev = torch.device("cuda:0")
sd = synthetic_state_dict(23, seed=0)
engines, streams, xs, ys = [], [], [], []
for i in range(a.engines):
    e = native.Engine(num_block=23, group=a.group, precision=a.hp)
    e.load_state_dict(sd)
    engines.append(e)
    streams.append(torch.cuda.Stream(device=dev))
    g = torch.Generator(device="cpu").manual_seed(100 + i)
    xs.append(torch.randint(0, 256, (a.batch, 256, 256, 3), dtype=torch.uint8, generator=g).to(dev))
    ys.append(torch.empty((a.batch, 1024, 1024, 3), dtype=torch.uint8, device=dev))
torch.cuda.synchronize()


def step():
    for e, s, x, y in zip(engines, streams, xs, ys):
        e.forward_batch_u8_dev(x.data_ptr(), a.batch, 256, 256, y.data_ptr(), s.cuda_stream)


for _ in range(a.warmup):
    step()
torch.cuda.synchronize()
cs = ClockSampler(0)
cs.start()
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
clk = cs.stop() or {}
tiles = a.engines * a.batch
print(f"engines={a.engines} batch={a.batch} group={a.group}: {dt * 1e3:8.2f} ms per {tiles} tiles  {tiles * 1.048576 / dt:7.1f} SR-MP/s  "
      f"{clk.get('sclk_mhz')} MHz {clk.get('power_w')} W", flush=True)
