#!/bin/bash
# group-size sweep of the whole net (fast and hp modes)
for hp in 0 1; do
  for g in 2 4 8 16 32; do
    python tools/quick_bench.py --batch 32 --steps 2 --group $g --prof 1 --hp $hp 2>&1 | grep -E "B=|rdb_conv"
  done
done
