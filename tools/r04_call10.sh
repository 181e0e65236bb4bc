#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/r04_c10; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_trunk.py -x -q -m gpu -k "f16" > $OUT/t1.log 2>&1
rc=$?; echo "[c10] trunk tests rc=$rc"; tail -3 $OUT/t1.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python3 -m pytest tests/test_gpu_net.py -x -q -m gpu -k "full_size_batch_properties or g4_full or hp_mode or graph_replay or batch_consistency or degenerate or ragged or whole_patch or real_image or g5 or shape_changes" > $OUT/t2.log 2>&1
rc=$?; echo "[c10] net tests rc=$rc"; tail -3 $OUT/t2.log
[ $rc -eq 0 ] || exit 1
bash tools/ab_latency.sh S2SR_SMALL_PL "1 2" > $OUT/ab_small_pl.txt 2>&1
grep -E "==|latency" $OUT/ab_small_pl.txt
