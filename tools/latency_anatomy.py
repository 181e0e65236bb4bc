#!/usr/bin/env python3
"""Where one tile's 4.5 ms go: is it the gaps BETWEEN the 351 dependent launches, or the launches?

Two modes:
  run   <S> [n]    -- n graph-replayed forwards of one SxS tile (HP mode); meant to sit behind `rocprofv3 --kernel-trace`
  sum   <dir>      -- read the kernel-trace csv under <dir>: per forward (a run of 353 dispatches ending in conv_last's
                      kernel) the wall time first-start -> last-end, the sum of kernel durations, the sum of gaps,
                      and the gap / duration per kernel family
"""
import csv
import glob
import re
import sys
from collections import defaultdict
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "sentinel2-super-resolution-poc_amd"))


def run(S: int, n: int):
    import torch
    from s2sr import native
    from s2sr.weights import synthetic_state_dict
    e = native.Engine(num_block=23, precision=native.PREC_F16_HP)
    e.load_state_dict(synthetic_state_dict(23, seed=0))
    x = torch.randint(0, 256, (1, S, S, 3), dtype=torch.uint8, device="cuda:0")
    y = torch.empty((1, 4 * S, 4 * S, 3), dtype=torch.uint8, device="cuda:0")
    side = torch.cuda.Stream()
    for _ in range(n + 3):
        e.forward_batch_u8_dev(x.data_ptr(), 1, S, S, y.data_ptr(), side.cuda_stream)
        torch.cuda.synchronize()
    print("graphs", e.graph_stats())
    e.close()


def fam(name: str) -> str:
    if "conv_trunk_f16<1" in name:
        return "rdb_conv1-4"
    if "conv_trunk_f16<2" in name:
        return "rdb_conv5"
    if "conv3x3" in name or "conv_phase" in name:
        return "head/tail conv"
    return "other"


def summarize(d: str):
    files = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    if not files:
        print("no kernel_trace.csv under", d)
        return
    rows = []
    for r in csv.DictReader(open(files[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # split into forwards: a forward ends with the EPI_LAST conv (the last conv3x3 instance before a pack kernel)
    fwd, cur = [], []
    for s, e, nme in rows:
        if "pack_u8" in nme and cur:
            fwd.append(cur)
            cur = []
        cur.append((s, e, nme))
    if cur:
        fwd.append(cur)
    fwd = [f for f in fwd if len(f) > 300]
    print(f"{len(fwd)} forwards of {len(fwd[-1])} dispatches in {files[0]}")
    for f in fwd[-3:]:
        wall = f[-1][1] - f[0][0]
        dur = sum(e - s for s, e, _ in f)
        gaps = [f[i + 1][0] - f[i][1] for i in range(len(f) - 1)]
        print(f"  wall {wall / 1e3:8.1f} us   sum(kernel) {dur / 1e3:8.1f} us   sum(gaps) {sum(gaps) / 1e3:7.1f} us   "
              f"mean gap {sum(gaps) / len(gaps) / 1e3:5.2f} us   max gap {max(gaps) / 1e3:5.2f} us")
    f = fwd[-1]
    per = defaultdict(lambda: [0, 0.0, 0.0])
    for i, (s, e, nme) in enumerate(f):
        k = per[fam(nme)]
        k[0] += 1
        k[1] += e - s
        if i + 1 < len(f):
            k[2] += f[i + 1][0] - e
    for k, (n, du, ga) in per.items():
        print(f"  {k:16s} launches {n:4d}  mean duration {du / n / 1e3:7.2f} us  mean gap after {ga / n / 1e3:5.2f} us")
    names = defaultdict(lambda: [0, 0.0])
    for s, e, nme in f:
        nm = re.sub(r"\(.*", "", nme.replace("s2sr::", "").replace("(anonymous namespace)::", "").replace("void ", ""))[:70]
        names[nm][0] += 1
        names[nm][1] += e - s
    for nm, (n, du) in sorted(names.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"    {nm:70s} x{n:4d}  {du / n / 1e3:7.2f} us each")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 5)
    else:
        summarize(sys.argv[2])
