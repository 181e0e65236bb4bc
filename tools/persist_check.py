import sys
sys.path.insert(0, "sentinel2-super-resolution-poc_amd")
from s2sr import native
e = native.Engine(num_block=1)
for variant in (4, 5, 4, 5):
    for grid, P in ((256, 2), (256, 4), (64, 2)):
        r = e.rdb_persistent(variant, grid, P, 23, 4)
        print(f"variant={variant} ({'device-scope loads, written-through stores' if variant == 5 else 'plain loads and stores'}) grid={grid} P={P}: "
              f"halo mismatches {r['halo_mismatches']}  own mismatches {r['own_mismatches']}  timeouts {r['timeouts']}  {r['TFLOP_per_s']:.0f} TFLOP/s", flush=True)
