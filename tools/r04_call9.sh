#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/r04_c9; mkdir -p $OUT
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 S2SR_FORCE_DIST=1 timeout -k 10 600 python3 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_forcedist.json 2> $OUT/bench_forcedist.err
echo "[c9] bench over RCCL (one rank, the N>1 code path) rc=$?"; tail -c 400 $OUT/bench_forcedist.err; python3 tools/show_bench.py $OUT/bench_forcedist.json | head -8
bash tools/r04_gpu_suite.sh r04_suite2
