#!/usr/bin/env python3
"""Do two builds of libs2sr.so give the same bytes?  Used for the short e4m3 encodings of the split-operand producers
(conv3x3.hip, S2SR_HPO_SHORT): build the long form next to the shipped library (tools/build_diag_libs.sh hpo0) and run
  python tools/check_hpo_forms.py sentinel2-super-resolution-poc_amd/csrc/diag/libs2sr_hpo0.so
Each library is loaded in its own process (S2SR_LIB); outputs of HP nets on the goldens' inputs, a stress net (body gain 1)
and noise scaled to exercise the e4m3 clamps are compared byte for byte."""
import os
import subprocess
import sys
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, "sentinel2-super-resolution-poc_amd")
from s2sr import native
from s2sr.weights import synthetic_state_dict
g = np.load("tests/golden/g4_full_nets.npz")
rng = np.random.default_rng(5)
out = {}
for nb, kw in ((6, {}), (23, {}), (23, {"body_gain": 1.0})):
    e = native.Engine(num_block=nb, precision=native.PREC_F16_HP)
    e.load_state_dict(synthetic_state_dict(nb, seed=0, **kw))
    out[f"g4_{nb}_{len(kw)}"] = e.forward_f32(g["x"])
    out[f"noise_{nb}_{len(kw)}"] = e.forward_f32(rng.random((2, 3, 45, 70), dtype=np.float32))
    out[f"big_{nb}_{len(kw)}"] = e.forward_f32((rng.random((1, 3, 33, 40), dtype=np.float32) * 40.0 - 20.0).astype(np.float32))
    out[f"u8_{nb}_{len(kw)}"] = e.forward_batch_u8(rng.integers(0, 256, (3, 64, 96, 3), dtype=np.uint8))
    e.close()
np.savez(sys.argv[1], **out)
'''


def run(lib, path):
    env = dict(os.environ)
    if lib:
        env["S2SR_LIB"] = str(Path(lib).resolve())
    subprocess.run([sys.executable, "-c", CHILD, path], check=True, cwd=REPO, env=env)


if __name__ == "__main__":
    import numpy as np
    first, other = (None, sys.argv[1]) if len(sys.argv) == 2 else (sys.argv[1], sys.argv[2])   # one argument: the shipped library vs it
    with tempfile.TemporaryDirectory() as d:
        run(first, d + "/a.npz"); run(other, d + "/b.npz")
        a, b = np.load(d + "/a.npz"), np.load(d + "/b.npz")
        bad = 0
        for k in a.files:
            same = np.array_equal(a[k], b[k], equal_nan=True)
            diff = 0.0 if same else float(np.nanmax(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))))
            print(f"{k:16s} {'identical' if same else 'DIFFERENT'}  max|d| {diff:.3e}  finite {bool(np.isfinite(a[k]).all())}")
            bad += not same
        sys.exit(1 if bad else 0)
