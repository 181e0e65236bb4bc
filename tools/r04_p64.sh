#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/r04_p64; mkdir -p $OUT
timeout -k 10 300 python3 -m pytest tests/test_gpu_trunk.py -x -q -m gpu -k "whole_patch" > $OUT/t1.log 2>&1
rc=$?; echo "[p64] per-layer rc=$rc"; tail -3 $OUT/t1.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2 3; do
  for v in 0 1; do
    echo "== S2SR_F16_P64=$v rep $rep"
    S2SR_F16_P64=$v timeout -k 10 200 python3 tools/quick_bench.py --batch 32 --steps 6 --hp 1 2>&1 | grep -E "B=|rdb_conv"
  done
done | tee $OUT/ab.txt
