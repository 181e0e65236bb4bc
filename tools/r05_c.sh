#!/bin/bash
# r05 third GPU call: D2H probe through the library, the tests touched since the last suite, the bench line with the ceiling sweep
R=$PWD; OUT=$R/gpurun_out/${1:-r05_c}; mkdir -p $OUT
timeout -k 10 200 python3 tools/d2h_probe.py > $OUT/d2h.txt 2>&1; echo "[r05_c] d2h rc=$?"; grep -v "^/opt" $OUT/d2h.txt
timeout -k 10 400 python3 -m pytest tests/test_gpu_net.py::test_cut_forward_stitch_equals_enhance tests/test_gpu_net.py::test_dist_aoi_chunked_equals_enhance tests/test_gpu_tiles.py -x -q -m gpu > $OUT/t1.log 2>&1
rc=$?; echo "[r05_c] tests rc=$rc"; tail -4 $OUT/t1.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 560 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
rc=$?; echo "[r05_c] bench rc=$rc"; tail -3 $OUT/bench.err; python3 tools/show_bench.py $OUT/bench.json | grep -E "SR-MP/s|aoi|mfma_ceiling"
