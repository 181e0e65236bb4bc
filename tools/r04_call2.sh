#!/bin/bash
# r04 second GPU call: the new dist / workspace / configs[3] tests, the CPU probe, the new bench line
R=$PWD
OUT=$R/gpurun_out/r04_c2
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_net.py -x -q -m gpu -k "dist_aoi or share_one_workspace or config3 or cut_forward_stitch or rccl_world1 or aoi_enhance_crops or window_mosaics or config2" > $OUT/t1.log 2>&1
echo "[c2] tests rc=$?"; tail -5 $OUT/t1.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_app.py -x -q -m gpu -k "two_ranks" > $OUT/t2.log 2>&1
echo "[c2] rehearsal rc=$?"; tail -5 $OUT/t2.log
timeout -k 10 300 python3 tools/cpu_probe.py > $OUT/cpu_probe.txt 2>&1
echo "[c2] probe rc=$?"; cat $OUT/cpu_probe.txt
timeout -k 10 600 python3 bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
echo "[c2] bench rc=$?"; tail -c 300 $OUT/bench.err
python3 tools/show_bench.py $OUT/bench.json 2>/dev/null | tail -40
