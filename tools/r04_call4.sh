#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/r04_c4; mkdir -p $OUT
timeout -k 10 300 python3 tools/launch_anatomy.py 256 256 > $OUT/launch_anatomy_256.txt 2>&1; echo "rc=$?"; cat $OUT/launch_anatomy_256.txt
timeout -k 10 300 python3 tools/launch_anatomy.py 64 64 > $OUT/launch_anatomy_64.txt 2>&1; echo "rc=$?"; cat $OUT/launch_anatomy_64.txt
