#!/bin/bash
# r05 second GPU call: launch-group sweep (is a MALL-resident group cheaper in power?), D2H probe, then the whole GPU suite
R=$PWD; OUT=$R/gpurun_out/${1:-r05_b}; mkdir -p $OUT
for g in 4 8 16 32; do
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --group $g > $OUT/bench_g$g.json 2> $OUT/bench_g$g.err
  echo "[r05_b] group $g rc=$?"; python3 tools/show_bench.py $OUT/bench_g$g.json | head -2
done
timeout -k 10 120 python3 tools/d2h_probe.py > $OUT/d2h.txt 2>&1; echo "[r05_b] d2h rc=$?"; cat $OUT/d2h.txt | grep -v "^/opt"
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $OUT/suite.log 2>&1
echo "[r05_b] suite rc=$?"; tail -4 $OUT/suite.log
