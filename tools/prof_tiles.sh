#!/bin/bash
# rocprofv3 kernel-trace stats of the XYZ pyramid stage (tools/bench_tiles.py): level kernels + the two PNG encoder kernels
#   gpurun -- 'bash tools/prof_tiles.sh'  -> gpurun_out/prof_tiles/
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_tiles; mkdir -p $OUT
python3 $GRAFT_REPO_ROOT/tools/bench_tiles.py 4096 > $OUT/bench_tiles.txt 2>&1; tail -4 $OUT/bench_tiles.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/bench_tiles.py 4096 > $OUT/trace.log 2>&1
f=$(ls $OUT/trace/*/*kernel_stats.csv | head -1); head -8 "$f" | cut -c1-220 | tee $OUT/kernel_stats_head.csv
