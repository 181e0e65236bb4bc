#!/usr/bin/env python3
"""8-wave conv kernel (conv3x3.hip) vs the one-wave-per-SIMD trunk kernel (conv_trunk.hip) on RDB-shaped convs,
sustained (>= 2 s of back-to-back launches per row so DVFS has settled): launch time, cycles per launch and the
in-kernel clock (s_memtime / s_memrealtime of the stamped build's last launch), socket power from hwmon.
    S2SR_TRACE_TIMED=1 python tools/trunk_anatomy.py"""
import glob
import os
import sys
import threading
import time
from pathlib import Path

import numpy as np

os.environ["S2SR_TRACE_TIMED"] = "1"
REPO = Path(__file__).resolve().parent.parent
os.environ.setdefault("S2SR_LIB", str(REPO / "sentinel2-super-resolution-poc_amd" / "csrc" / "libs2sr_exp.so"))   # stamped builds: make -C csrc EXP=1
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))
import torch  # noqa: E402,F401
from s2sr import native  # noqa: E402


def power_files():
    try:
        pr = torch.cuda.get_device_properties(0)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        hw = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
        return (hw[0] + "/power1_input", hw[0] + "/freq1_input") if hw else None
    except Exception:
        return None


class Sampler:
    def __init__(self):
        self.f, self.w, self.c, self.stop = power_files(), [], [], threading.Event()

    def run(self):
        while self.f and not self.stop.wait(0.05):
            try:
                self.w.append(int(open(self.f[0]).read()) / 1e6)
                self.c.append(int(open(self.f[1]).read()) / 1e6)
            except Exception:
                return


engines = {}
for name, env in (("8-wave", "0"), ("trunk", "1")):
    os.environ["S2SR_TRUNK"] = env
    engines[name] = native.Engine(num_block=1)
N, H, W = 16, 256, 256
print(f"{'kernel':8s} {'cin':>4s} {'cout':>4s} {'dbg':>3s} {'us/launch':>10s} {'TFLOP/s':>8s} {'kcycles':>8s} {'clock GHz':>9s} {'W':>6s} {'hwmon MHz':>9s}")
for cin, cout in ((160, 32), (64, 32), (192, 64)):
    for name in ("8-wave", "trunk", "8-wave", "trunk"):
        for dbg in ((0, 8) if name == "trunk" else (0,)):
            os.environ["S2SR_DBG"] = str(dbg)
            fl = 2.0 * N * H * W * cin * 9 * cout
            iters = int(2.2e6 / (fl / 0.85e9))          # ~2.2 s at 0.85 PFLOP/s
            smp = Sampler()
            th = threading.Thread(target=smp.run, daemon=True)
            th.start()
            us, tr = engines[name].bench_conv(N, H, W, cin, cout, iters=iters, trace_wgs=256)
            smp.stop.set()
            th.join(1)
            tr = tr.astype(np.int64)
            tr = tr[tr[:, 20] > 0]
            cyc = np.median(tr[:, 23] - tr[:, 22])
            clk = np.median((tr[:, 23] - tr[:, 22]) / np.maximum(tr[:, 21] - tr[:, 20], 1)) * 0.1
            tail = slice(len(smp.w) // 2, None)
            pw = np.mean(smp.w[tail]) if smp.w else float("nan")
            hc = np.mean(smp.c[tail]) if smp.c else float("nan")
            print(f"{name:8s} {cin:4d} {cout:4d} {dbg:3d} {us:10.1f} {fl / us / 1e6:8.0f} {cyc / 1e3:8.1f} {clk:9.2f} {pw:6.0f} {hc:9.0f}", flush=True)
