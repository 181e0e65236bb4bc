#!/bin/bash
# usage: tools/prof_pmc.sh <name>   -> gpurun_out/<name>/{trace,pmc_*}: rocprofv3 kernel trace of bench.py
# plus separate PMC passes (one counter group per pass, as MI355X_MICROARCH.md prescribes) on a
# small run of the same hot path.
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "[prof] kernel trace of bench.py"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1
echo "[prof] trace done rc=$?"
ARGS="$R/tools/quick_bench.py --batch 8 --steps 1 --prof 0 --group 8 --hp 1"
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  echo "[prof] pmc $tag"
  timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$tag -- python3 $ARGS > $OUT/pmc_$tag.log 2>&1
  echo "[prof] pmc $tag rc=$?"
done
cd $R
python3 tools/summarize_prof.py gpurun_out/$1 > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +6M -delete
echo "[prof] done"
