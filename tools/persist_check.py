#!/usr/bin/env python3
"""Check of the cross-XCD hand-over of the persistent-loop prototype (csrc/persist.hip variants 4 / 5): the plane stores carry (layer count, writer)
and every landed halo / own piece is compared -- plain loads and stores against device-scope loads and written-through stores
(profiles/r05_persistent_loop.txt)."""
import sys
sys.path.insert(0, "sentinel2-super-resolution-poc_amd")
from s2sr import native
e = native.Engine(num_block=1)
for variant in (4, 5, 4, 5):
    for grid, P in ((256, 2), (256, 4), (64, 2)):
        r = e.rdb_persistent(variant, grid, P, 23, 4)
        print(f"variant={variant} ({'device-scope loads, written-through stores' if variant == 5 else 'plain loads and stores'}) grid={grid} P={P}: "
              f"halo mismatches {r['halo_mismatches']}  own mismatches {r['own_mismatches']}  timeouts {r['timeouts']}  {r['TFLOP_per_s']:.0f} TFLOP/s", flush=True)
