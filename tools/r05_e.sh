#!/bin/bash
# r05 fifth GPU call: post-process band kernels timed, ceiling sweep incl. the kernel's own traffic mix
R=$PWD; OUT=$R/gpurun_out/${1:-r05_e}; mkdir -p $OUT
timeout -k 10 300 python3 tools/pp_band_probe.py > $OUT/pp.txt 2>&1; echo "[r05_e] pp probe rc=$?"; grep -v "^/opt" $OUT/pp.txt
timeout -k 10 300 python3 - > $OUT/ceil.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, "sentinel2-super-resolution-poc_amd"); sys.path.insert(0, ".")
import bench
from s2sr import native
e = native.Engine(num_block=1)
import json
print(json.dumps(bench.mfma_ceiling_leg(e, 0), indent=1))
PY
echo "[r05_e] ceiling rc=$?"; grep -v "^/opt" $OUT/ceil.txt | grep -E "TFLOP_per_s|sclk|power|\"[a-z_]+\": \{" 
