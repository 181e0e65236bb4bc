#!/usr/bin/env python3
"""Latency of whole-image forwards of upload-sized images (one launch sequence, 23 blocks, HP) for the patch-form thresholds given in the environment."""
import os
import sys
import time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from s2sr import native  # noqa: E402
from s2sr.weights import synthetic_state_dict  # noqa: E402
e = native.Engine(num_block=23, precision=native.PREC_F16_HP)
e.load_state_dict(synthetic_state_dict(23, seed=0))
rng = np.random.default_rng(0)
out = []
for (h, w) in ((256, 256), (320, 320), (432, 576), (512, 512), (384, 640), (200, 1000)):
    x = torch.from_numpy(rng.integers(0, 256, (1, h, w, 3), dtype=np.uint8)).cuda()
    y = torch.empty((1, 4 * h, 4 * w, 3), dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(4):
            e.forward_batch_u8_dev(x.data_ptr(), 1, h, w, y.data_ptr(), s.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            e.forward_batch_u8_dev(x.data_ptr(), 1, h, w, y.data_ptr(), s.cuda_stream)
        torch.cuda.synchronize()
    out.append(f"{h}x{w}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms")
print(" ".join(f"{k}={os.environ[k]}" for k in sorted(os.environ) if k.startswith("S2SR_PROBE")) or "default", "|", "  ".join(out))
