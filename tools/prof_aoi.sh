#!/bin/bash
# kernel-trace of the AOI path (tools/bench_aoi.py 4096): is the wall time kernels or gaps?
OUT=$PWD/gpurun_out/$1
R=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export S2SR_AOI_TILES=256
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/bench_aoi.py 4096 > $OUT/trace.log 2>&1
echo "rc=$?"
cd $R
grep -v amdgpu.ids $OUT/trace.log | tail -3
f=$(ls $OUT/trace/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time total {tot/1e6:.1f} ms over all enhance() calls")
for r in rows[:12]:
    print(f"  {r['Name'][:90]:90s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:9.1f} {r['Percentage']}%")
PY
find $OUT -name "*.csv" -size +4M -delete
