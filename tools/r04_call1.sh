#!/bin/bash
# r04 first GPU call: baseline bench at HEAD + single-tile launch-gap anatomy (rocprofv3 kernel trace)
R=$PWD
OUT=$R/gpurun_out/r04_c1
mkdir -p $OUT
python3 bench.py --steps 10 --warmup 3 > $OUT/bench_base.json 2> $OUT/bench_base.err
echo "[c1] bench rc=$?"
cd /tmp && export TMPDIR=/tmp
for S in 256 64; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/lat_$S -- python3 $R/tools/latency_anatomy.py run $S 5 > $OUT/lat_$S.log 2>&1
  echo "[c1] trace $S rc=$?"
  python3 $R/tools/latency_anatomy.py sum $OUT/lat_$S > $OUT/lat_${S}_summary.txt 2>&1
done
cd $R
find $OUT -name "*.csv" -size +8M -delete
find $OUT -name "*.db" -delete
tail -c 600 $OUT/bench_base.json; cat $OUT/lat_256_summary.txt $OUT/lat_64_summary.txt
