#!/bin/bash
# finer launch-group sweep (is there a sweet spot between MALL residency and per-launch overhead?)
R=$PWD; OUT=$R/gpurun_out/${1:-r05_m}; mkdir -p $OUT
for cfg in "16 32" "10 30" "12 36" "14 28" "20 40" "24 48" "16 32"; do
  set -- $cfg
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --group $1 --batch $2 > $OUT/bench_g$1.json 2> $OUT/bench_g$1.err
  python3 - $OUT/bench_g$1.json $1 $2 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]; f = r["families"]; c = r.get("held_clock", {})
print(f"G={sys.argv[2]:>2s} batch={sys.argv[3]:>2s}: {d['value']:7.2f} SR-MP/s  conv1-4 {f['rdb_conv1-4']['TFLOP_per_s']:7.1f}  conv5 {f['rdb_conv5']['TFLOP_per_s']:7.1f} TF/s  up {f['conv_up']['TFLOP_per_s']} hr {f['conv_hr']['TFLOP_per_s']}  {c.get('sclk_mhz')} MHz {c.get('power_w')} W")
PY
done
