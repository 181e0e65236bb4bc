#!/usr/bin/env python3
"""Device-memory stability over a few hundred mixed calls (enhance / forward / post-process, changing shapes)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / 'sentinel2-super-resolution-poc_amd'))
import numpy as np, torch
from s2sr import native
from s2sr.weights import synthetic_state_dict
e=native.Engine(num_block=2, precision=native.PREC_F16_HP); e.load_state_dict(synthetic_state_dict(2,seed=0))
rng=np.random.default_rng(0)
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]/2**20
sizes=[(24,32),(40,56),(64,64),(33,70),(100,90),(17,200)]
for s in sizes: e.enhance_u8(rng.integers(0,256,(*s,3),dtype=np.uint8), tile=16, pad=2)
f0=free()
for it in range(300):
    s=sizes[it%len(sizes)]
    img=rng.integers(0,256,(*s,3),dtype=np.uint8)
    if it%3==0: e.enhance_u8(img, tile=16, pad=2)
    elif it%3==1: e.forward_batch_u8(img[None])
    else: e.postprocess_u8(np.repeat(np.repeat(img,2,0),2,1), native.pp_wow())
    if it%100==99: print(it, f"free MiB {free():.0f} (start {f0:.0f}) graphs {e.graph_stats()}")
e.close(); print("after close free", f"{free():.0f}")

# ... and the host side: page-locked outputs (native.pinned_pool) and the staged device-to-host bands must not grow the process
import gc
import psutil
e = native.Engine(num_block=1, precision=native.PREC_F16_HP); e.load_state_dict(synthetic_state_dict(1, seed=0))
proc = psutil.Process()
big = rng.integers(0, 256, (1300, 1200, 3), dtype=np.uint8)
for pinned in (True, False):
    native.pinned_pool.on = pinned
    for _ in range(3):
        e.enhance_u8(big, tile=256, pad=10)
    gc.collect()
    r0 = proc.memory_info().rss / 2**20
    for it in range(30):
        out = None
        out = e.enhance_u8(big, tile=256, pad=10)
        if it % 7 == 0:
            keep = out[:10].copy()
    out = None
    gc.collect()
    r1 = proc.memory_info().rss / 2**20
    print(f"pinned outputs {pinned}: host RSS {r0:.0f} -> {r1:.0f} MiB over 30 calls of a 75-MB result; pool hits {native.pinned_pool.hits} misses {native.pinned_pool.misses}")
    assert r1 - r0 < 200, "host memory grows"
native.pinned_pool.on = True
native.pinned_pool.trim()
e.close()
