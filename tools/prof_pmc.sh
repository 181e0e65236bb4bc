#!/bin/bash
# usage: tools/prof_pmc.sh <outdir-under-gpurun_out> -- runs kernel-trace stats + PMC passes on quick_bench
set -e
OUT=gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="$R/tools/quick_bench.py --batch 8 --steps 1 --prof 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace -- python3 $ARGS > $R/$OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/$OUT/pmc1 -- python3 $ARGS > $R/$OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $R/$OUT/pmc2 -- python3 $ARGS > $R/$OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/$OUT/pmc3 -- python3 $ARGS > $R/$OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc4 -- python3 $ARGS > $R/$OUT/pmc4.log 2>&1
cd $R
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1 || true
# keep only small files
find $OUT -name "*.csv" -size +8M -delete
