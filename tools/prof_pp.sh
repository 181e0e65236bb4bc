#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/pp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/bench_pp.py > $OUT/trace.log 2>&1
echo rc=$?
cd $GRAFT_REPO_ROOT
tail -3 $OUT/trace.log
python3 tools/summarize_prof.py gpurun_out/pp | head -12
find gpurun_out/pp -name "*.csv" -size +6M -delete
