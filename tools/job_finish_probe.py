#!/usr/bin/env python3
"""Where the enhance_crops flavour of a 4096x4096 AOI spends the time behind its last window (S2SR_JOB_TIMING=1 prints the finish's
stage times from inside s2sr_enhance_job_u8): plain job against enhance_crops job, page-locked output."""
import os
import sys
import time
from pathlib import Path

os.environ["S2SR_JOB_TIMING"] = "1"
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
import numpy as np  # noqa: E402

from s2sr import native  # noqa: E402
from s2sr.synth import synthetic_tiles  # noqa: E402
from s2sr.weights import synthetic_state_dict  # noqa: E402

nb = int(sys.argv[2]) if len(sys.argv) > 2 else 23
side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
e = native.Engine(num_block=nb, precision=native.PREC_F16_HP)
e.load_state_dict(synthetic_state_dict(nb, seed=0))
img = synthetic_tiles(1, side, seed=4321)[0]
for prm, name in ((None, "plain"), (native.pp_wow(), "enhance_crops")):
    for rep in range(4):
        out = None
        t0 = time.perf_counter()
        out = e.enhance_job_u8(img, prm)
        dt = time.perf_counter() - t0
        print(f"{name} rep {rep}: {dt * 1e3:.1f} ms", flush=True)
