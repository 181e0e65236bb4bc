#!/bin/bash
# usage (on the GPU box, from the repo root): GIT_REV=<rev> tools/prof_pmc.sh <name>
#   -> gpurun_out/<name>/: rocprofv3 kernel-trace stats of `bench.py` (hp and fp8 lines) + separate PMC passes
#      (one counter group per pass, as MI355X_MICROARCH.md prescribes) on one group of 16 tiles per mode,
#      then tools/summarize_prof.py -> summary.txt and pmc_summary.json (stamped with GIT_REV: the box has no .git, so the
#      caller passes `git rev-parse --short HEAD` -- tools/gpu_prof.sh does)
export GIT_REV=${GIT_REV:-unknown}
OUT=$PWD/gpurun_out/$1
R=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in hp fp8; do
  echo "[prof] kernel trace of bench.py --precision $mode"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$mode -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --precision $mode > $OUT/trace_$mode.log 2>&1
  echo "[prof] trace $mode rc=$?"
done
for mode in hp fp8; do
  hpflag=1; [ $mode = fp8 ] && hpflag=2
  ARGS="$R/tools/quick_bench.py --batch 16 --steps 1 --prof 0 --group 16 --hp $hpflag"
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_WAVES" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $grp | cut -d' ' -f1)
    echo "[prof] pmc $mode $tag"
    timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_${mode}_$tag -- python3 $ARGS > $OUT/pmc_${mode}_$tag.log 2>&1
    echo "[prof] pmc $mode $tag rc=$?"
  done
done
cd $R
python3 tools/summarize_prof.py gpurun_out/$1 > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +4M -delete
find $OUT -name "*.db" -delete
echo "[prof] done"
