#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/r04_final; mkdir -p $OUT
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "[final] smoke rc=$?"; tail -1 $OUT/smoke.txt
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_hp.json 2> $OUT/bench_hp.err; echo "[final] bench rc=$?"
timeout -k 10 600 python3 bench.py --steps 100 --warmup 10 --no-secondary --no-cpu-baseline > $OUT/bench_sustained.json 2>/dev/null; echo "[final] sustained rc=$?"
python3 tools/show_bench.py $OUT/bench_hp.json $OUT/bench_sustained.json | grep -v "cpu leg"
timeout -k 10 300 python3 tools/bench_latency.py > $OUT/latency.txt 2>&1; grep "blocks=23" $OUT/latency.txt
