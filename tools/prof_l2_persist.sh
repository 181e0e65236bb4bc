#!/bin/bash
# Does L2 keep lines across dependent kernels?  FETCH_SIZE per image at group 1 (3.3 MB working set per XCD) vs group 8.
OUT=$GRAFT_REPO_ROOT/gpurun_out/l2p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for b in 1 8; do
  echo "[l2p] batch $b"
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/b$b -- python3 $R/tools/quick_bench.py --batch $b --group $b --blocks 6 --steps 1 --prof 0 > $OUT/b$b.log 2>&1
  echo "[l2p] rc=$?"
done
cd $R
for b in 1 8; do mkdir -p gpurun_out/l2p_$b; rm -rf gpurun_out/l2p_$b/pmc_F; cp -r $OUT/b$b gpurun_out/l2p_$b/pmc_F; python3 tools/summarize_prof.py gpurun_out/l2p_$b | grep -A2 "conv3x3_f16<1\|conv3x3_f16<2, 2, 8, 1\|conv3x3_f16<2, 2, 8, 2" ; done
find gpurun_out -name "*.csv" -size +6M -delete
