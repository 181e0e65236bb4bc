#!/usr/bin/env python3
"""fp8 trunk: error against the 23-block golden as a function of the activation-scale exponents."""
import os
import sys
from pathlib import Path
import numpy as np
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
from s2sr import native  # noqa: E402
from s2sr.weights import synthetic_state_dict  # noqa: E402
g4 = np.load(REPO / "tests" / "golden" / "g4_full_nets.npz")
sd = synthetic_state_dict(23, seed=0)
sd1 = synthetic_state_dict(23, seed=0, body_gain=1.0)
for xe in (1, 2, 3, 4, 5, 6, 7):
    for ge in (3, 4, 5, 6, 7, 8, 9):
        os.environ["S2SR_FP8_XEXP"], os.environ["S2SR_FP8_GEXP"] = str(xe), str(ge)
        out = []
        for s, key in ((sd, "y_b23"), (sd1, "y_b23_gain1")):
            e = native.Engine(num_block=23, precision=native.PREC_FP8)
            e.load_state_dict(s)
            d = np.abs(e.forward_f32(g4["x"]) - g4[key])
            out.append((d.max(), np.sqrt((d ** 2).mean())))
            e.close()
        print(f"x_exp {xe} g_exp {ge}: max-abs {out[0][0]:.3e} rms {out[0][1]:.3e} | stress max-abs {out[1][0]:.3e} rms {out[1][1]:.3e}", flush=True)
