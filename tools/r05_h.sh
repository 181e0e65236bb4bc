#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/${1:-r05_h}; mkdir -p $OUT
timeout -k 10 500 python3 -m pytest tests/test_gpu_tiles.py tests/test_gpu_app.py -x -q -m gpu > $OUT/t1.log 2>&1
rc=$?; echo "[r05_h] tests rc=$rc"; tail -4 $OUT/t1.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2; do
S2SR_PNG_TIMING=1 timeout -k 10 300 python3 tools/bench_job.py 1024 > $OUT/job$rep.txt 2>&1; echo "[r05_h] bench_job rc=$?"; grep -E "process_wow_sr 1024|read GeoTIFF|SR net|write GeoTIFF|other|process_raster_to_tiles|read [0-9]|9801 tiles|2500 tiles| 650 tiles" $OUT/job$rep.txt
done
