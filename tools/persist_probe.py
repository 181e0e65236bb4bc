#!/usr/bin/env python3
"""Probe: the RDB-shaped loop whose workgroups stay across layers (csrc/persist.hip through s2sr_debug_rdb_persistent) over working sets inside
and beyond the Infinity Cache, with and without device-scope loads / write-through stores; clock and socket power sampled meanwhile.

    python3 tools/persist_probe.py [--rdbs 23] [--launches 8] [--quick]
"""
import argparse
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
sys.path.insert(0, str(REPO))
from bench import ClockSampler  # noqa: E402
from s2sr import native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rdbs", type=int, default=23)
ap.add_argument("--launches", type=int, default=8)
ap.add_argument("--quick", action="store_true")
a = ap.parse_args()
e = native.Engine(num_block=1)
cases = [(256, 2)] if a.quick else [(256, 2), (128, 2), (256, 4)]
for coherent in ((0, 2) if a.quick else (1, 3, 0, 2)):      # variant: bit 0 coherent, bit 1 deeper ring
    for grid, P in cases:
        r = e.rdb_persistent(coherent, grid, P, 2 if a.quick else a.rdbs, 1 if a.quick else 2)          # settle; a hang would show here on a short run
        if r["timeouts"]:
            print(f"variant={coherent} grid={grid} P={P}: {r['timeouts']} dependency waits ran into their bound -- not timed", flush=True)
            continue
        cs = ClockSampler(0)
        cs.start()
        r = e.rdb_persistent(coherent, grid, P, 2 if a.quick else a.rdbs, 1 if a.quick else a.launches)
        clk = cs.stop() or {}
        print(f"variant={coherent} grid={grid:3d} P={P} working set {r['working_set_MB']:6.1f} MB: {r['TFLOP_per_s']:7.1f} TFLOP/s "
              f"(x 256 / grid: {r['TFLOP_per_s'] * 256 / grid:7.1f})  {r['ms'] / r['launches']:7.2f} ms per launch  timeouts {r['timeouts']}  "
              f"{clk.get('sclk_mhz')} MHz {clk.get('power_w')} W", flush=True)
