#!/bin/bash
# r04 A/B: fp16 conv1-4 with the weights from global memory (WGL) against the shipped form, same box
R=$PWD; OUT=$R/gpurun_out/r04_wgl; mkdir -p $OUT
timeout -k 10 300 python3 -m pytest tests/test_gpu_trunk.py -x -q -m gpu -k "whole_patch" > $OUT/t1.log 2>&1
rc=$?; echo "[wgl] per-layer rc=$rc"; tail -3 $OUT/t1.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2 3; do
  for v in 0 1; do
    echo "== S2SR_F16_WGL=$v rep $rep"
    S2SR_F16_WGL=$v timeout -k 10 200 python3 tools/quick_bench.py --batch 32 --steps 6 --hp 1 2>&1 | grep -E "B=|rdb_conv"
  done
done | tee $OUT/ab.txt
