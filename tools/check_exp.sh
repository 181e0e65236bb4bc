#!/bin/bash
# The experimental library (make EXP=1): build it and run the static asm checks on its sources, conv_wino.hip included.
#   tools/check_exp.sh            -> csrc/libs2sr_exp.so
#   S2SR_LIB=$PWD/sentinel2-super-resolution-poc_amd/csrc/libs2sr_exp.so python -m pytest tests -m gpu     (on a GPU box: also runs the tests marked `experimental`)
set -e
C=sentinel2-super-resolution-poc_amd/csrc
make -C $C EXP=1 -j8 > /dev/null
T=$(mktemp -d)
for f in conv_trunk conv3x3 conv_wino; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DS2SR_EXPERIMENTAL=1 -S --cuda-device-only $C/$f.hip -o $T/$f.s 2>/dev/null
  python3 tools/check_asm_loads.py $T/$f.s | tail -1
done
rm -rf $T
