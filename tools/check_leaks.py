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
