#!/bin/bash
# one-wave-per-SIMD (S2SR_W4=1) forms of the RDB convs: correctness first, then timing, then the per-wave stage anatomy
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== correctness (W4): conv + net goldens" 
S2SR_W4=1 timeout -k 10 300 python -m pytest tests/test_gpu_net.py -x -q -m gpu -k "g3 or g4 or hp_mode or batch_consistency or full_size_batch or tiled_vs_oracle" > gpurun_out/w4_tests.log 2>&1 || { tail -30 gpurun_out/w4_tests.log; exit 1; }
tail -3 gpurun_out/w4_tests.log
echo "== timing 8-wave" && timeout -k 10 200 python tools/quick_bench.py --batch 32 --steps 3 --hp 1 2>&1 | tee gpurun_out/w4_qb8.log
echo "== timing W4" && S2SR_W4=1 timeout -k 10 200 python tools/quick_bench.py --batch 32 --steps 3 --hp 1 2>&1 | tee gpurun_out/w4_qb4.log
echo "== wave trace 8-wave" && timeout -k 10 200 python tools/trace_waves.py 2>&1 | tee gpurun_out/w4_trace8.log
echo "== wave trace W4" && S2SR_W4=1 timeout -k 10 200 python tools/trace_waves.py 2>&1 | tee gpurun_out/w4_trace4.log
