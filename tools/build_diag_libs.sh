#!/bin/bash
# Diagnostic builds of libs2sr.so for the split-operand tail convs (conv3x3.hip), made in the build container next to the
# shipped library (csrc/diag/*.so travel to the GPU box with the snapshot; they are git-ignored):
#   libs2sr_f8diag{1,2,4,8}.so  -DS2SR_DIAG_F8=n   1 no LDS-DMA, 2 no MFMA, 4 no epilogue, 8 epilogue without stores (tools/tail_anatomy.sh)
#   libs2sr_f8diag16.so         diagnostics hooks only (S2SR_DIAG_GRID=<workgroups>)
#   libs2sr_hpo0.so             -DS2SR_HPO_SHORT=0     long form of the lo e4m3 encoding (tools/check_hpo_forms.py)
# Usage: tools/build_diag_libs.sh [names...]   (default: all);  rm -rf csrc/diag afterwards.
set -e
C=sentinel2-super-resolution-poc_amd/csrc
make -C $C > /dev/null
mkdir -p $C/diag
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -Wno-unused-value -ffp-contract=off -DS2SR_EXPERIMENTAL=0"
OBJS="engine.o conv_trunk.o pack.o postprocess.o hostcodec.o tiles.o"
build() {  # name, define
  /opt/rocm/bin/hipcc $FLAGS $2 -c $C/conv3x3.hip -o $C/diag/conv3x3_$1.o
  (cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o diag/libs2sr_$1.so $OBJS diag/conv3x3_$1.o)
  rm -f $C/diag/conv3x3_$1.o
  echo built $C/diag/libs2sr_$1.so
}
want=${@:-f8diag1 f8diag2 f8diag4 f8diag8 f8diag16 hpo0}
for n in $want; do
  case $n in
    f8diag*) build $n "-DS2SR_DIAG_F8=${n#f8diag}" & ;;
    hpo0) build hpo0 "-DS2SR_HPO_SHORT=0" & ;;
    *) echo "unknown $n"; exit 1 ;;
  esac
done
wait
