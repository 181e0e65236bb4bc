#!/usr/bin/env python3
"""One tile's latency (23 blocks, HP, graph replays): 64x64 and 256x256, best of 3 runs of 30 calls."""
import sys
import time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
import torch  # noqa: E402
from s2sr import native  # noqa: E402
from s2sr.weights import synthetic_state_dict  # noqa: E402
e = native.Engine(num_block=23, precision=native.PREC_F16_HP)
e.load_state_dict(synthetic_state_dict(23, seed=0))
side = torch.cuda.Stream()
out = []
for S in (64, 256):
    x = torch.randint(0, 256, (1, S, S, 3), dtype=torch.uint8, device="cuda:0")
    y = torch.empty((1, 4 * S, 4 * S, 3), dtype=torch.uint8, device="cuda:0")
    for _ in range(4):
        e.forward_batch_u8_dev(x.data_ptr(), 1, S, S, y.data_ptr(), side.cuda_stream)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(30):
            e.forward_batch_u8_dev(x.data_ptr(), 1, S, S, y.data_ptr(), side.cuda_stream)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 30)
    out.append(f"{S}x{S}: {best * 1e3:.3f} ms")
print("latency  " + "   ".join(out), flush=True)
e.close()
