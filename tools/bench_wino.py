#!/usr/bin/env python3
"""A/B on one box: the whole net with RDB conv1-4 direct (conv_trunk_f16) vs row-Winograd F(2,3) (conv_wino.hip, S2SR_WINO=1),
HP mode, 32 tiles per step, per-family HIP-event stats, A/B/A/B."""
import os, subprocess, sys
for rep in (1, 2):
    for w in ("0", "1", "2"):
        env = dict(os.environ, S2SR_WINO=w)
        print(f"== S2SR_WINO={w} rep {rep}", flush=True)
        out = subprocess.run([sys.executable, "tools/quick_bench.py", "--batch", "32", "--steps", "4", "--hp", "1"], env=env, capture_output=True, text=True, timeout=300)
        print("\n".join(l for l in out.stdout.splitlines() if "B=" in l or "rdb_conv" in l), flush=True)
        if out.returncode:
            print(out.stderr[-2000:])
