#!/bin/bash
# what do the row-Winograd form's 4 packed adds per MFMA cost?  conv_wino.hip with / without its transform instructions
# (S2SR_WINO_DIAG_NOXFORM=1: wrong results, timing only), whole net with S2SR_WINO=1, A/B/A/B on one box
set -o pipefail
C=sentinel2-super-resolution-poc_amd/csrc
for rep in 1 2; do
for v in 0 1; do
  rm -f $C/conv_wino.o
  make -C $C CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-value -DS2SR_WINO_DIAG_NOXFORM=$v" > /dev/null 2>&1 || { echo "build failed"; exit 1; }
  echo "== S2SR_WINO_DIAG_NOXFORM=$v rep $rep"
  S2SR_WINO=1 timeout -k 10 200 python tools/quick_bench.py --batch 32 --steps 4 --hp 1 2>&1 | grep -E "B=|rdb_conv"
done
done
rm -f $C/conv_wino.o; make -C $C > /dev/null 2>&1
