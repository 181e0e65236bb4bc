#!/usr/bin/env python3
"""Post-process (CLAHE + unsharp + vegetation) throughput on device-resident 1024x1024 RGB tiles."""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
import torch
from s2sr import native
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
e = native.Engine(num_block=1)
dev = torch.device("cuda:0")
x = torch.randint(0, 256, (B, 1024, 1024, 3), dtype=torch.uint8, device=dev)
y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
for prm, name in ((native.pp_wow(), "wow"), (native.pp_farm(), "farm")):
    e.postprocess_batch_u8_dev(x.data_ptr(), B, 1024, 1024, prm, y.data_ptr(), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        e.postprocess_batch_u8_dev(x.data_ptr(), B, 1024, 1024, prm, y.data_ptr(), st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    px = B * 1024 * 1024
    print(f"{name}: B={B} {dt*1e3:.2f} ms  {px/dt/1e6:.0f} MP/s  {px*15/dt/1e9:.0f} GB/s at 15 B/px (this version's traffic), "
          f"{px*9/dt/1e9:.0f} GB/s at the 9 B/px floor")
