#!/bin/bash
# A/B of a conv_trunk.hip compile-time switch on ONE box, single-tile latency and the 32-tile step: tools/ab_latency.sh MACRO "v1 v2"
M=$1; VALS=$2
C=sentinel2-super-resolution-poc_amd/csrc
# the variants are built over the shipped objects: put the shipped library back on ANY exit (r03 ADVICE: an interrupted run left a diagnostic build where the tests load it)
trap 'rm -f $C/conv_trunk.o; make -C $C > /dev/null 2>&1' EXIT
for rep in 1 2; do
for v in $VALS; do
  rm -f $C/conv_trunk.o
  make -C $C CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-value -D$M=$v" > /dev/null 2>&1 || { echo "build failed for $M=$v"; exit 1; }
  echo "== $M=$v rep $rep"
  timeout -k 10 200 python3 tools/lat_quick.py 2>&1 | grep latency
  timeout -k 10 200 python3 tools/quick_bench.py --batch 32 --steps 4 --hp 1 2>&1 | grep -E "B=|rdb_conv"
done
done
