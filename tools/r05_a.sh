#!/bin/bash
# r05 first GPU call: the banded post-process tests, then the bench line with its new legs
R=$PWD; OUT=$R/gpurun_out/${1:-r05_a}; mkdir -p $OUT
timeout -k 10 500 python3 -m pytest tests/test_gpu_postprocess.py tests/test_gpu_app.py::test_enhance_job_equals_the_separate_steps tests/test_gpu_net.py::test_dist_aoi_chunked_equals_enhance tests/test_gpu_net.py::test_aoi_enhance_crops_composition_vs_oracles -x -q -m gpu -s > $OUT/t1.log 2>&1
rc=$?; echo "[r05_a] tests rc=$rc"; tail -15 $OUT/t1.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 560 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
rc=$?; echo "[r05_a] bench rc=$rc"; tail -5 $OUT/bench.err; python3 tools/show_bench.py $OUT/bench.json | head -60
