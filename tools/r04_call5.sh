#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/r04_c5; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_trunk.py -x -q -m gpu -k "f16" > $OUT/t1.log 2>&1
echo "[c5] trunk tests rc=$?"; tail -3 $OUT/t1.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_net.py -x -q -m gpu -k "full_size_batch_properties or g4_full or hp_mode or graph_replay or batch_consistency or degenerate or ragged or whole_patch or window_mosaics or real_image" > $OUT/t2.log 2>&1
echo "[c5] net tests rc=$?"; tail -3 $OUT/t2.log
timeout -k 10 300 python3 tools/bench_latency.py > $OUT/latency.txt 2>&1
echo "[c5] latency rc=$?"; grep "blocks=23" $OUT/latency.txt
timeout -k 10 300 python3 tools/launch_anatomy.py 256 256 > $OUT/launch_anatomy_256.txt 2>&1; cat $OUT/launch_anatomy_256.txt
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
echo "[c5] bench rc=$?"; python3 tools/show_bench.py $OUT/bench.json | head -12
