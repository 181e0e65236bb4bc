#!/bin/bash
# group-size / workgroup-shape sweep on one GPU
mkdir -p gpurun_out
for g in 2 4 8 16 32; do
  python tools/quick_bench.py --batch 32 --steps 2 --group $g --prof 0 2>&1 | grep "B="
done
echo "--- CT1 4 waves (2 WG/CU)"
for g in 4 8 16; do
  S2SR_CT1_WAVES=4 python tools/quick_bench.py --batch 32 --steps 2 --group $g --prof 1 2>&1 | grep -E "B=|rdb_conv"
done
