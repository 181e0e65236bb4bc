#!/usr/bin/env python3
"""Condensed view of bench.py JSON lines."""
import json
import sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        hv = r.get("hbm_view", {})
        print(f"{f}: {d['value']} {d['unit']}  {d['ms_per_step']} ms/step  dtype {d['dtype']}  dominant {r['kernel']} bound {r['bound']} {r['achieved']} {r['unit']} "
              f"frac {r['frac']} (of fed ceiling {r.get('frac_of_fed_ceiling')}, mfma_busy {r.get('mfma_busy')})  rdb {r['rdb_convs_TFLOP_per_s']} ({r['rdb_convs_frac']})  hbm view {hv.get('achieved')} GB/s ({hv.get('frac')})  "
              f"timed {r['timed_pass']}  clock {r.get('held_clock')}")
        print("   " + "  ".join(f"{k}: {v['ms']} ms {v['TFLOP_per_s']} TF/s" for k, v in r["families"].items()))
        s = d.get("secondary", {})
        if "fp8" in s:
            print(f"   fp8: {s['fp8']['value']} SR-MP/s {s['fp8']['ms_per_step']} ms/step frac {s['fp8']['roofline']['frac']}")
        if "aoi" in s:
            print(f"   aoi: {s['aoi']['value']} SR-MP/s in {s['aoi']['seconds']} s (ideal {s['aoi']['ideal_at_batch_rate_s']} s); 1 tile {s.get('latency_ms_1tile')} ms")
        for k in ("aoi_tile512", "aoi_job_plain", "aoi_job_enhance_crops", "aoi_1024", "aoi_dist_world1", "enhance_crops_b64", "job_1024", "ref_recorded_job", "mfma_ceiling"):
            if k in s:
                print(f"   {k}: " + "  ".join(f"{kk}={vv}" for kk, vv in s[k].items() if kk != "workload"))
        if "latency_ms_64x64" in s:
            print(f"   64x64 latency {s['latency_ms_64x64']} ms")
        if "aoi_strong_scaling" in d:
            print("   aoi_strong_scaling: " + "  ".join(f"{kk}={vv}" for kk, vv in d["aoi_strong_scaling"].items() if kk not in ("workload", "check")))
        if "cpu_baseline" in d:
            for k, v in d["cpu_baseline"].get("legs", {}).items():
                print(f"   cpu leg {k}: {v}")
        if "cpu_baseline" in d:
            c = d["cpu_baseline"]
            print(f"   cpu: {c['value']} SR-MP/s on {c['cores']} threads of {c.get('cpu_model')}; one thread {c.get('one_thread', {}).get('value')}")
    except Exception as e:
        print(f, "ERR", e)
