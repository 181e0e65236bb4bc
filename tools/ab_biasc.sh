#!/bin/bash
# A/B on one box: conv_trunk_f16 conv1-4 with the bias in the MFMA's C operand (default) vs added in the epilogue
set -o pipefail
C=sentinel2-super-resolution-poc_amd/csrc
for rep in 1 2; do
for v in 1 0; do
  rm -f $C/conv_trunk.o
  make -C $C CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-value -DS2SR_F16_BIASC=$v" > /dev/null 2>&1 || exit 1
  echo "== S2SR_F16_BIASC=$v rep $rep"
  timeout -k 10 200 python tools/quick_bench.py --batch 32 --steps 4 --hp 1 2>&1 | grep -E "B=|rdb_conv"
done
done
rm -f $C/conv_trunk.o; make -C $C > /dev/null 2>&1
