#!/bin/bash
# the one-wave-per-SIMD trunk kernel (conv_trunk.hip, default on; S2SR_TRUNK=0 = the 8-wave kernel): correctness, timing, per-wave anatomy
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-all}
echo "== correctness" 
timeout -k 10 400 python -m pytest tests/test_gpu_net.py tests/test_gpu_conv.py -x -q -m gpu -k "not config2 and not full_size_configs" > gpurun_out/trunk_tests.log 2>&1 || { tail -40 gpurun_out/trunk_tests.log; exit 1; }
tail -3 gpurun_out/trunk_tests.log
echo "== timing 8-wave" && S2SR_TRUNK=0 timeout -k 10 200 python tools/quick_bench.py --batch 32 --steps 3 --hp 1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/trunk_qb8.log
echo "== timing trunk" && timeout -k 10 200 python tools/quick_bench.py --batch 32 --steps 3 --hp 1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/trunk_qb4.log
echo "== wave trace trunk" && timeout -k 10 200 python tools/trace_waves.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/trunk_trace4.log
