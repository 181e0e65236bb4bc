#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/${1:-r05_f}; mkdir -p $OUT
timeout -k 10 300 python3 -m pytest tests/test_gpu_postprocess.py -x -q -m gpu > $OUT/t1.log 2>&1; rc=$?; echo "[r05_f] pp tests rc=$rc"; tail -3 $OUT/t1.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 tools/pp_band_probe.py > $OUT/pp.txt 2>&1; echo "[r05_f] pp probe rc=$?"; grep -v "^/opt" $OUT/pp.txt
timeout -k 10 300 python3 tools/job_finish_probe.py 4096 > $OUT/finish.txt 2>&1; echo "[r05_f] finish probe rc=$?"; grep -v "^/opt" $OUT/finish.txt | tail -12
timeout -k 10 300 python3 - > $OUT/ceil.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, "sentinel2-super-resolution-poc_amd"); sys.path.insert(0, ".")
import bench
from s2sr import native
e = native.Engine(num_block=1)
import json
print(json.dumps(bench.mfma_ceiling_leg(e, 0), indent=1))
PY
echo "[r05_f] ceiling rc=$?"; grep -v "^/opt" $OUT/ceil.txt | grep -E "TFLOP_per_s|sclk|power|lds_dma_GB|\"[a-z_]+\": \{" | paste - - - - - | head
