#!/bin/bash
# probe: groups of 8 images (dense tensor of an RDB = 200 MB: MALL-resident) on 16x32 patches (1024 patches per launch = 4 per workgroup)
R=$PWD; OUT=$R/gpurun_out/${1:-r05_i}; mkdir -p $OUT
for cfg in "16 192" "8 192" "8 600" "4 300" "16 1100"; do
  set -- $cfg
  S2SR_PROBE_N32_16=$2 timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --group $1 > $OUT/bench_g$1_t$2.json 2> $OUT/bench_g$1_t$2.err
  echo "[r05_i] group $1 threshold $2 rc=$?"; python3 tools/show_bench.py $OUT/bench_g$1_t$2.json | head -2 | cut -c1-700
done
