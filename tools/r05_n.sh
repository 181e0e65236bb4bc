#!/bin/bash
# engines side by side on their own streams, small launch groups (tools/two_stream_probe.py)
R=$PWD; OUT=$R/gpurun_out/${1:-r05_n}; mkdir -p $OUT
for cfg in "1 32 16" "2 16 16" "2 16 8" "2 16 4" "4 8 4" "3 12 4" "4 8 8" "2 16 2" "4 8 2" "1 32 16"; do
  set -- $cfg
  timeout -k 10 120 python3 tools/two_stream_probe.py --engines $1 --batch $2 --group $3 --steps 10 2>> $OUT/err.log | tee -a $OUT/probe.txt
done
