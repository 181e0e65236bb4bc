#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/${1:-r05_l}; mkdir -p $OUT
for cfg in "" "S2SR_PROBE_N32_16=700 S2SR_PROBE_N16_5=192" "S2SR_PROBE_N32_16=700 S2SR_PROBE_N16_5=1400" "S2SR_PROBE_N32_8=700 S2SR_PROBE_N16_5=1400" "S2SR_PROBE_N32_8=300 S2SR_PROBE_N32_16=700 S2SR_PROBE_N16_5=600" ""; do
  env $cfg timeout -k 10 200 python3 tools/form_probe.py 2>&1 | grep -v "^/opt" | tee -a $OUT/forms.txt
done
