#!/bin/bash
# conv5's traffic mix as a seventh ceiling loop: the test of the entry, then the loops twice with clock and power
R=$PWD; OUT=$R/gpurun_out/${1:-r05_w}; mkdir -p $OUT
timeout -k 10 300 python3 -m pytest tests/test_gpu_net.py -x -q -m gpu -k "mfma_ceiling" > $OUT/t.log 2>&1; echo "test rc=$?"; tail -3 $OUT/t.log
timeout -k 10 400 python3 - > $OUT/ceil.txt 2> $OUT/ceil.err <<'PY'
import sys
sys.path.insert(0, "sentinel2-super-resolution-poc_amd"); sys.path.insert(0, ".")
import bench
from s2sr import native
from s2sr.weights import synthetic_state_dict
e = native.Engine(num_block=1)
for rep in range(2):
    d = bench.mfma_ceiling_leg(e, 0)
    for k, v in d.items():
        if isinstance(v, dict):
            print(f"{k:28s} {v['TFLOP_per_s']:7.1f} TFLOP/s  {v.get('sclk_mhz')} MHz {v.get('power_w')} W  dma {v.get('lds_dma_GB_per_s')}", flush=True)
PY
echo "ceil rc=$?"; cat $OUT/ceil.txt; tail -3 $OUT/ceil.err
