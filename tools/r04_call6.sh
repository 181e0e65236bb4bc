#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/r04_c6; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_trunk.py -x -q -m gpu -k "f16" > $OUT/t1.log 2>&1
echo "[c6] trunk tests rc=$?"; tail -3 $OUT/t1.log
bash tools/ab_latency.sh S2SR_F16_TAIL "0 1" > $OUT/ab_tail.txt 2>&1
cat $OUT/ab_tail.txt
