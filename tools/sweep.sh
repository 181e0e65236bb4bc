#!/bin/bash
for g in 2 4 8 16 32; do
  python tools/quick_bench.py --batch 32 --steps 2 --group $g --prof 1 2>&1 | grep -E "B=|rdb_conv"
done
echo "--- 4 waves x 4 rows"
for g in 4 8; do
  S2SR_WAVES=4 python tools/quick_bench.py --batch 32 --steps 2 --group $g --prof 1 2>&1 | grep -E "B=|rdb_conv"
done
