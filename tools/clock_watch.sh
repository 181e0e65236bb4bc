#!/bin/bash
# Sample sclk / power with rocm-smi while a steady bench loop runs (diagnostic for DVFS limits).
mkdir -p gpurun_out
python tools/quick_bench.py --steps 60 "$@" > gpurun_out/cw_bench.log 2>&1 &
BP=$!
for i in $(seq 1 40); do
  sleep 0.5
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' >> gpurun_out/cw_smi.log
  echo >> gpurun_out/cw_smi.log
  kill -0 $BP 2>/dev/null || break
done
wait $BP
tail -3 gpurun_out/cw_bench.log
tail -25 gpurun_out/cw_smi.log
