#!/bin/bash
# r05 fourth GPU call: tile / app tests after the PNG-stage host changes, stage timings of the pyramid and of the job finish
R=$PWD; OUT=$R/gpurun_out/${1:-r05_d}; mkdir -p $OUT
timeout -k 10 500 python3 -m pytest tests/test_gpu_tiles.py tests/test_gpu_app.py -x -q -m gpu > $OUT/t1.log 2>&1
rc=$?; echo "[r05_d] tests rc=$rc"; tail -4 $OUT/t1.log
[ $rc -eq 0 ] || exit 1
S2SR_PNG_TIMING=1 timeout -k 10 300 python3 tools/bench_job.py 1024 > $OUT/job.txt 2>&1; echo "[r05_d] bench_job rc=$?"; grep -v "^/opt\|Loaded\|Device" $OUT/job.txt | tail -40
timeout -k 10 300 python3 tools/job_finish_probe.py 4096 > $OUT/finish.txt 2>&1; echo "[r05_d] finish probe rc=$?"; grep -v "^/opt" $OUT/finish.txt | tail -20
