#!/usr/bin/env python3
"""A/B on one box: fp16 conv1-4 with / without the load-only fifth wave (S2SR_F16_LOADER), whole net, HP, 32 tiles per step."""
import os, subprocess, sys
for rep in (1, 2):
    for w in ("0", "1"):
        env = dict(os.environ, S2SR_F16_LOADER=w)
        print(f"== S2SR_F16_LOADER={w} rep {rep}", flush=True)
        out = subprocess.run([sys.executable, "tools/quick_bench.py", "--batch", "32", "--steps", "4", "--hp", "1"], env=env, capture_output=True, text=True, timeout=300)
        print("\n".join(l for l in out.stdout.splitlines() if "B=" in l or "rdb_conv" in l), flush=True)
        if out.returncode:
            print(out.stderr[-2000:])
