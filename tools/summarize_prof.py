#!/usr/bin/env python3
"""Summarise rocprofv3 csv output of tools/prof_pmc.sh per kernel: kernel-trace stats per mode, PMC sums per
dispatch, and profiles-ready pmc_summary.json (HBM bytes per image and launch of the RDB conv kernels:
FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md + WRITE_SIZE, both reported in KiB)."""
import csv
import glob
import json
import os
import re
import subprocess
import sys
from collections import defaultdict

out = sys.argv[1]
IMGS = 16          # images per launch in the PMC runs (tools/prof_pmc.sh: --batch 16 --group 16)


def short(n):
    n = n.replace("s2sr::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(s2sr::ConvParams\)|\(ConvParams\)", "", n)[:74]


for mode in ("hp", "fp8"):
    for f in glob.glob(f"{out}/trace_{mode}/**/*kernel_stats.csv", recursive=True):
        print(f"== kernel stats ({mode})", f)
        for r in list(csv.DictReader(open(f)))[:16]:
            print(f"  {short(r['Name']):74s} calls {r['Calls']:>6s} total_ns {r['TotalDurationNs']:>12s} avg_ns {r['AverageNs']:>10s} pct {r['Percentage']}")

pmc = {}
for mode in ("hp", "fp8"):
    per = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for p in sorted(glob.glob(f"{out}/pmc_{mode}_*/")):
        for f in glob.glob(f"{p}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                c = per[k][r["Counter_Name"]]
                c[0] += float(r["Counter_Value"])
                c[1] += 1
    print(f"== pmc ({mode}), per dispatch")
    for k in sorted(per, key=lambda k: -per[k].get("SQ_WAVE_CYCLES", [0, 1])[0])[:8]:
        print(" ", k)
        for c, (v, n) in sorted(per[k].items()):
            print(f"      {c:28s} per-dispatch {v / max(n, 1):16.0f}  (n={n})")
    pmc[mode] = per

try:
    rev = os.environ.get("GIT_REV") or subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except Exception:
    rev = None
summary = {"_meta": {"group": IMGS, "git_rev": rev, "tool": "tools/prof_pmc.sh",
                     "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes on tools/quick_bench.py --batch 16 --group 16; "
                             "FETCH_SIZE doubled (gfx950 tallies 128-B requests as 64 B); counter units KiB"}}
FAM = {"hp": {"rdb_conv1-4": ("conv_trunk_f16<1, 8, 3, 0", 18874368), "rdb_conv5": ("conv_trunk_f16<2, 4, 4, 1", 41943040),
              "rdb_conv5_rrdb": ("conv_trunk_f16<2, 4, 4, 2", 54525952)},
       "fp8": {"rdb_conv1-4": ("conv_trunk_f8<1, 4, 6, 0", 9437184), "rdb_conv5": ("conv_trunk_f8<2, 4, 4, 1", 33554432),
               "rdb_conv5_rrdb": ("conv_trunk_f8<2, 4, 4, 2", 41943040)}}
for mode, fams in FAM.items():
    for fam, (pat, alg) in fams.items():
        # a family can span several template instances (fp8 conv1-3 keep weights resident, conv4 streams them):
        # pool their counters so the per-launch mean covers the same launches as the algorithmic mean
        ks = [k for k, cs in pmc.get(mode, {}).items() if pat in k and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs]
        if not ks:
            continue
        fs = [sum(pmc[mode][k][c][i] for k in ks) for c in ("FETCH_SIZE", "WRITE_SIZE") for i in (0, 1)]
        fk, wk = fs[0] / fs[1], fs[2] / fs[3]
        ent = {
            "kernel": " + ".join(sorted(ks)), "dispatches": fs[1], "images_per_launch": IMGS, "fetch_kib_raw": round(fk),
            "write_kib": round(wk), "hbm_bytes_per_image": round((2 * fk + wk) * 1024 / IMGS),
            "algorithmic_bytes_per_image_mean": alg}
        # matrix-pipe utilisation: SQ_VALU_MFMA_BUSY_CYCLES (summed over the SIMDs) / (1024 SIMDs x the kernel's cycles); the
        # kernel's cycles = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 -- the two counters come from different passes
        def mean(c):
            v = [pmc[mode][k][c] for k in ks if c in pmc[mode][k]]
            return sum(x[0] for x in v) / max(sum(x[1] for x in v), 1) if v else None
        busy, gui = mean("SQ_VALU_MFMA_BUSY_CYCLES"), mean("GRBM_GUI_ACTIVE")
        if busy and gui:
            ent["mfma_busy"] = round(busy / (1024 * gui / 8), 4)
            ent["mfma_busy_note"] = "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), per dispatch means of separate passes"
        summary[("" if mode == "hp" else "fp8:") + fam] = ent
json.dump(summary, open(f"{out}/pmc_summary.json", "w"), indent=1)
print("== pmc_summary.json")
print(json.dumps(summary, indent=1))
