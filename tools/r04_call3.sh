#!/bin/bash
# r04 third GPU call: conv5 8x32-patch form -- per-layer parity, byte identity with the batch form, single-tile latency
R=$PWD
OUT=$R/gpurun_out/r04_c3
mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_trunk.py -x -q -m gpu -k "conv5" > $OUT/t1.log 2>&1
echo "[c3] trunk tests rc=$?"; tail -5 $OUT/t1.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_net.py -x -q -m gpu -k "full_size_batch_properties or g4_full or hp_mode or graph_replay or batch_consistency or degenerate" > $OUT/t2.log 2>&1
echo "[c3] net tests rc=$?"; tail -5 $OUT/t2.log
timeout -k 10 300 python3 tools/bench_latency.py > $OUT/latency.txt 2>&1
echo "[c3] latency rc=$?"; cat $OUT/latency.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/lat_256 -- python3 $R/tools/latency_anatomy.py run 256 5 > $OUT/lat_256.log 2>&1
python3 $R/tools/latency_anatomy.py sum $OUT/lat_256 > $OUT/lat_256_summary.txt 2>&1
cat $OUT/lat_256_summary.txt
cd $R; find $OUT -name "*.csv" -size +8M -delete; find $OUT -name "*.db" -delete
