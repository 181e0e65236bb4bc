#!/bin/bash
# r05: tile / app tests after the host-side changes (arena, native XYZ paths, literal LZW), job + pyramid stage timings, then the
# tests of the buried kernel forms on the experimental library
R=$PWD; OUT=$R/gpurun_out/${1:-r05_g}; mkdir -p $OUT
timeout -k 10 500 python3 -m pytest tests/test_gpu_tiles.py tests/test_gpu_app.py -x -q -m gpu > $OUT/t1.log 2>&1
rc=$?; echo "[r05_g] tests rc=$rc"; tail -4 $OUT/t1.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2; do
S2SR_PNG_TIMING=1 timeout -k 10 300 python3 tools/bench_job.py 1024 > $OUT/job$rep.txt 2>&1; echo "[r05_g] bench_job rc=$?"; grep -E "process_wow_sr 1024|read GeoTIFF|SR net|write GeoTIFF|other|process_raster_to_tiles|read [0-9]|9801 tiles|2500 tiles" $OUT/job$rep.txt
done
S2SR_LIB=$R/sentinel2-super-resolution-poc_amd/csrc/libs2sr_exp.so timeout -k 10 900 python3 -m pytest tests -x -q -m "gpu and experimental" > $OUT/suite_exp.log 2>&1
echo "[r05_g] experimental library rc=$?"; tail -4 $OUT/suite_exp.log
