#!/usr/bin/env python3
"""What the host of a GPU box gives the CPU baseline: logical / physical cores, cgroup quota, and how one RDB of the oracle
(the unit 92 % of the net is made of) scales with oneDNN threads at batch 1 and at batch 4."""
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from oracle import rrdbnet_ref as ref  # noqa: E402
from s2sr.weights import synthetic_state_dict  # noqa: E402

print(bench.host_cpu_facts(), bench.cpu_model(), flush=True)
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/devices/system/cpu/smt/active"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, e)
print("loadavg", open("/proc/loadavg").read().strip(), flush=True)
sd = ref.to_torch_sd(synthetic_state_dict(1, seed=0))
flop = 2 * 9 * (64 * 32 + 96 * 32 + 128 * 32 + 160 * 32 + 192 * 64) * 256 * 256
for nb in (1, 4):
    x = torch.rand(nb, 64, 256, 256)
    for n in (1, 2, 4, 8, 16, 32, 64, 128):
        if n > (os.cpu_count() or 1):
            break
        torch.set_num_threads(n)
        with torch.no_grad():
            ref.rdb_forward(x, sd, "body.0.rdb1")
            t0 = time.perf_counter()
            reps = 2 if n < 4 else 4
            for _ in range(reps):
                ref.rdb_forward(x, sd, "body.0.rdb1")
            dt = (time.perf_counter() - t0) / reps
        print(f"batch {nb} threads {n:3d}: {dt * 1e3:8.1f} ms per RDB  {nb * flop / dt / 1e9:8.1f} GFLOP/s", flush=True)
