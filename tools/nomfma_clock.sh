#!/bin/bash
# bench.py lines (with held clock / power) of the S2SR_DIAG_NOMFMA=1 build, then the normal build restored
set -o pipefail
C=sentinel2-super-resolution-poc_amd/csrc
rm -f $C/conv_trunk.o
make -C $C CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-value -DS2SR_DIAG_NOMFMA=1" > /dev/null 2>&1 || { echo build failed; exit 1; }
python bench.py --no-cpu-baseline > gpurun_out/nm_hp.json 2>/dev/null
python bench.py --no-cpu-baseline --precision fp8 > gpurun_out/nm_fp8.json 2>/dev/null
rm -f $C/conv_trunk.o; make -C $C > /dev/null 2>&1
python tools/show_bench.py gpurun_out/nm_hp.json gpurun_out/nm_fp8.json
