#!/usr/bin/env python3
"""Probe: how much of the persistent loop's rate survives when the kernel covers only a few RDBs per launch (csrc/persist.hip: 69 per-RDB launches
instead of one per forward would keep the layer hand-over inside the kernel and the launch boundary between RDBs)."""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
sys.path.insert(0, str(REPO))
from bench import ClockSampler  # noqa: E402
from s2sr import native  # noqa: E402
e = native.Engine(num_block=1)
for P in (2, 4):
    for rdbs in (1, 2, 4, 8, 23, 69):
        launches = max(4, int(2400 / (rdbs * P)))
        e.rdb_persistent(1, 256, P, rdbs, max(2, launches // 8))
        cs = ClockSampler(0)
        cs.start()
        r = e.rdb_persistent(1, 256, P, rdbs, launches)
        clk = cs.stop() or {}
        print(f"P={P} rdbs per launch {rdbs:3d}: {r['TFLOP_per_s']:7.1f} TFLOP/s  {r['ms'] / launches * 1e3:9.1f} us per launch  timeouts {r['timeouts']}  "
              f"{clk.get('sclk_mhz')} MHz {clk.get('power_w')} W", flush=True)
