#!/usr/bin/env python3
"""Latency of small calls (launch-bound regime): B tiles of SxS, device-resident, HP mode."""
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
import torch  # noqa: E402

from s2sr import native  # noqa: E402
from s2sr.weights import synthetic_state_dict  # noqa: E402

for nb in (23, 6):
    e = native.Engine(num_block=nb, precision=native.PREC_F16_HP)
    e.load_state_dict(synthetic_state_dict(nb, seed=0))
    for B, S in ((1, 64), (1, 128), (1, 256), (2, 256), (4, 256), (8, 256), (16, 256)):
        x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, device="cuda:0")
        y = torch.empty((B, 4 * S, 4 * S, 3), dtype=torch.uint8, device="cuda:0")
        side = torch.cuda.Stream()      # the null stream cannot be graph-captured
        st = side.cuda_stream
        for _ in range(3):
            e.forward_batch_u8_dev(x.data_ptr(), B, S, S, y.data_ptr(), st)
        torch.cuda.synchronize()
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            e.forward_batch_u8_dev(x.data_ptr(), B, S, S, y.data_ptr(), st)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        cap, rep = e.graph_stats()
        print(f"[graphs {cap}/{rep}] blocks={nb} B={B} {S}x{S}: {1e3*(t2-t0)/n:.3f} ms/call (host enqueue {1e3*(t1-t0)/n:.3f} ms), "
              f"{B*16*S*S/1e6/((t2-t0)/n):.1f} SR-MP/s", flush=True)
    e.close()
