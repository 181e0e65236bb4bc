#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/r04_c7; mkdir -p $OUT
bash tools/ab_latency.sh S2SR_F16_EARLYBIAS "0 1" > $OUT/ab_earlybias.txt 2>&1
cat $OUT/ab_earlybias.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_trunk.py -x -q -m gpu -k "f16" > $OUT/t1.log 2>&1
echo "[c7] trunk tests rc=$?"; tail -3 $OUT/t1.log
