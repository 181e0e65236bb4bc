#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/r04_c8; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_app.py tests/test_gpu_net.py -x -q -m gpu -k "enhance_job or process_wow or page_locked or http or realesrgan_class or farm" > $OUT/t1.log 2>&1
echo "[c8] tests rc=$?"; tail -4 $OUT/t1.log
timeout -k 10 300 python3 tools/bench_job.py 1024 > $OUT/job_1024.txt 2>&1; cat $OUT/job_1024.txt | grep -v amdgpu.ids
timeout -k 10 300 python3 tools/check_leaks.py > $OUT/leaks.txt 2>&1; tail -4 $OUT/leaks.txt
