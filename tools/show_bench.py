#!/usr/bin/env python3
"""Condensed view of bench.py JSON lines."""
import json
import sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f"{f}: {d['value']} {d['unit']}  {d['ms_per_step']} ms/step  dtype {d['dtype']}  dominant {r['kernel']} bound {r['bound']} {r['achieved']} {r['unit']} "
              f"frac {r['frac']}  mfma {r.get('mfma')}  rdb {r['rdb_convs_TFLOP_per_s']}  hbm-ceiling {r['hbm_ceiling_TFLOP_per_s']}  timed {r['timed_pass']}  "
              f"clock {r.get('held_clock')}")
        print("   " + "  ".join(f"{k}: {v['ms']} ms {v['TFLOP_per_s']} TF/s" for k, v in r["families"].items()))
        if "cpu_baseline" in d:
            print("   cpu:", d["cpu_baseline"])
    except Exception as e:
        print(f, "ERR", e)
