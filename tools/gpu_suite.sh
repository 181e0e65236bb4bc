#!/bin/bash
# the whole GPU suite on the shipped library, then the tests of the buried forms on the experimental one (when built), then the bench line
R=$PWD; OUT=$R/gpurun_out/${1:-suite}; mkdir -p $OUT
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $OUT/suite.log 2>&1
rc=$?; echo "[suite] shipped library rc=$rc"; tail -4 $OUT/suite.log
[ $rc -eq 0 ] || exit 1
if [ -f sentinel2-super-resolution-poc_amd/csrc/libs2sr_exp.so ] && [ "${2:-}" = exp ]; then
  S2SR_LIB=$R/sentinel2-super-resolution-poc_amd/csrc/libs2sr_exp.so timeout -k 10 900 python3 -m pytest tests -x -q -m "gpu and experimental" > $OUT/suite_exp.log 2>&1
  echo "[suite] experimental library rc=$?"; tail -4 $OUT/suite_exp.log
fi
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "[suite] bench rc=$?"; python3 tools/show_bench.py $OUT/bench.json | cut -c1-900
