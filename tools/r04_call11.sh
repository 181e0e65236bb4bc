#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/r04_c11; mkdir -p $OUT
export S2SR_LIB=$R/sentinel2-super-resolution-poc_amd/csrc/libs2sr_exp.so
for rep in 1 2; do for v in 0 1; do echo "== exp lib S2SR_F16_WGL=$v rep $rep"; S2SR_F16_WGL=$v timeout -k 10 200 python3 tools/lat_quick.py 2>&1 | grep latency; done; done | tee $OUT/wgl_small.txt
